for T in 4 8 16 32 48; do
  echo "T=$T"; FH_COOP_T=$T timeout 200 python bench.py --no-cpu-baseline --steps 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'], d['roofline']['per_ray'])"
done
