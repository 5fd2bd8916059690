# describe_kernel with the BRIEF patch read from the RAW level (SNK_ORB_DESC_FAKE=2: the rows its moment window reads anyway) -- the memory side
# of "blur inside describe_kernel" (VERDICT round 3, item 3): timing only, descriptors are then meaningless.  1 = both windows contiguous.
B="python bench.py --no-cpu-baseline --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --check-frames 0 --frame-calls 0 --track-frames 0 --steps 10 --warmup 3"
for rep in 1 2; do for f in 0 2 1; do echo "== SNK_ORB_DESC_FAKE=$f"; SNK_ORB_DESC_FAKE=$f $B 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'))"; done; done
