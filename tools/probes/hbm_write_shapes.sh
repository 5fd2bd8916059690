#!/bin/bash
# Runs on the GPU box: write bandwidth of tools/probes/hbm_counter_probe's strip shapes (a wavefront writes `lanes` dwords per row at stride * strip).
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o t -- ${GRAFT_REPO_ROOT:-/root/repo}/tools/probes/hbm_counter_probe > /tmp/hp.log 2>&1
python - <<'PY'
import csv, re
bands = 32768
for r in csv.DictReader(open("/tmp/hp/t_kernel_stats.csv")):
    m = re.search(r"write_dword<(\d+), (\d+), (\d+)>", r["Name"])
    if m:
        strips, lanes, stride = map(int, m.groups())
        by = bands * 64 * strips * lanes * 4
        ns = float(r["MinNs"])
        print("strips %d x %2d lanes, stride %3d bytes: %5.0f MB in %6.1f us = %.2f TB/s" % (strips, lanes, stride, by / 1e6, ns / 1e3, by / ns / 1e3))
PY
