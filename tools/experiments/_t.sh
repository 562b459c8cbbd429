P='import json,sys; d=json.loads(sys.stdin.read()); h=d["hbm_gb"]; print(sys.argv[1], d["value"], d["ms_per_step"], "ms; device allocs in the timed region", h["device_allocs_in_timed_region"], "growth GB", h["reserved_growth_in_timed_region"], "reserved", h["allocator_peak_reserved"])'
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "$P" "fp32 default (warmup 2, 10 steps):"
done
python bench.py --no-cpu-baseline --no-roofline --no-extras --warmup 3 --steps 20 2>/dev/null | python -c "$P" "fp32 warmup 3, 20 steps:"
python bench.py --no-cpu-baseline --no-roofline --no-extras --warmup 1 --steps 20 2>/dev/null | python -c "$P" "fp32 warmup 1, 20 steps:"
python bench.py --dtype f16 --workload c5 --two-streams --no-cpu-baseline --no-roofline --no-extras --steps 5 --warmup 1 2>/dev/null | python -c "$P" "c5 two streams warmup 1:"
python bench.py --dtype f16 --workload c5 --two-streams --no-cpu-baseline --no-roofline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "$P" "c5 two streams warmup 2, 10 steps:"
python bench.py --dtype f16 --workload c5 --no-cpu-baseline --no-roofline --no-extras --steps 5 --warmup 1 2>/dev/null | python -c "$P" "c5 one stream warmup 1:"
python bench.py --clips 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "$P" "fp32 one clip per step:"
