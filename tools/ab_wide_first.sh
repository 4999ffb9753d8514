# usage (GPU box): bash tools/ab_wide_first.sh -- scan8m with the file numbering (off) against wide rows numbered first, same box
# (the "mesh" / "both" lines differed by a packTileLayout knob that the committed library no longer has: rings are always ordered)
set -o pipefail
b() { timeout -k 10 250 python bench.py --workload scan8m --steps 700 --warmup 7 --no-cpu-baseline --no-verify --no-variants --sustained-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f G  %.1f us' % (d['value']/1e9, 1e3*d['roofline']['avg_launch_ms']), d['config']['tiles'])"; }
for round in 1 2; do
  echo "round $round off   $(FEM_WIDE_FIRST=0 b)" || exit 1
  echo "round $round mesh  $(FEM_WIDE_FIRST=1 b)" || exit 1
  echo "round $round both  $(FEM_WIDE_FIRST=1 b)" || exit 1
done
timeout -k 10 250 python bench.py --workload scan8m --steps 100 --warmup 5 --no-cpu-baseline --no-variants --sustained-steps 0 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('verify', d['verify'])"
