mkdir -p gpurun_out/r03_v5; cd /tmp; export TMPDIR=/tmp
for v in base; do
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o f -- python3 /root/repo/tools/profile_c4_sweep.py --compact > /root/repo/gpurun_out/r03_v5/c4_$v.txt 2>&1
  cp /tmp/prof_$v/f_kernel_stats.csv /root/repo/gpurun_out/r03_v5/c4_${v}_kernel_stats.csv
  echo $v; grep "sweep ms" /root/repo/gpurun_out/r03_v5/c4_$v.txt; grep "k_render_fwd\|k_sort_tiles" /tmp/prof_$v/f_kernel_stats.csv | sed 's/(.*)",/ /'
done
cd /root/repo && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-c4 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
python -m pytest tests/test_parity_gpu.py tests/test_rotate360_fixture.py tests/test_round3_gpu.py -m gpu -x -q > gpurun_out/r03_v5/pytest_small_path.log 2>&1; tail -3 gpurun_out/r03_v5/pytest_small_path.log
