#!/bin/bash
# Same-box scan of SNK_BA_FILL_CHUNKS (chunks the fill pass of a 1024-window hand-over runs and uploads in): warm hand-over times.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for r in 1 2; do LIST=${1:-"4 8 16 2"}; for c in $LIST; do
  echo -n "chunks=$c: "; SNK_BA_FILL_CHUNKS=$c timeout 120 python tools/ba_batch_only.py --windows 1024 --creates 6 --solves 0 2>&1 | grep warm | tail -4 | awk '{printf "%s ", $3}'; echo
done; done
