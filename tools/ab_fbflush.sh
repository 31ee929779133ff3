for v in 1 0 1 0; do
  python -c "from kmers_amd import build; build.build(force=True, extra=['-DKMX_BS_FB_FLUSH=$v'])" >/dev/null 2>&1
  for k in 31 63; do
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline -k $k 2>/dev/null | python tools/bench_line.py flush=$v,k=$k
  done
done
