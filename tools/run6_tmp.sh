export TMPDIR=/tmp; mkdir -p gpurun_out/r3
line() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']; print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'launch', r['avg_launch_ms'], r.get('level'), d['config'].get('views_per_call'), d['phase_ms_per_step'], d.get('pool_placement',{}).get('tries'))"; }
timeout -k 10 800 python3 -m pytest tests -m gpu -q -x > gpurun_out/r3/gputest3.log 2>&1; echo "pytest rc $?"; tail -n 6 gpurun_out/r3/gputest3.log
timeout -k 10 200 python3 bench.py --no-cpu-baseline > gpurun_out/r3/b_c30.log 2>&1; line gpurun_out/r3/b_c30.log R2c30
timeout -k 10 200 python3 bench.py --no-cpu-baseline --chunk 64 --pool 64 > gpurun_out/r3/b_c60.log 2>&1; line gpurun_out/r3/b_c60.log R2c60
timeout -k 10 300 python3 bench.py --no-cpu-baseline --chunk 100 --pool 100 > gpurun_out/r3/b_c100.log 2>&1; line gpurun_out/r3/b_c100.log R2c100
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload R1 > gpurun_out/r3/b_r1.log 2>&1; line gpurun_out/r3/b_r1.log R1
VOXPROJ_LIB=$PWD/tools/libvoxproj_g4.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload R1 > gpurun_out/r3/b_r1_g4.log 2>&1; line gpurun_out/r3/b_r1_g4.log R1g4
VOXPROJ_LIB=$PWD/tools/libvoxproj_g4.so timeout -k 10 200 python3 bench.py --no-cpu-baseline > gpurun_out/r3/b_c30_g4.log 2>&1; line gpurun_out/r3/b_c30_g4.log R2c30g4
for mc in 1 2 4; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --views 38 --min-calls $mc > gpurun_out/r3/b_v38_mc$mc.log 2>&1; line gpurun_out/r3/b_v38_mc$mc.log v38mc$mc; done
timeout -k 10 300 python3 tools/bench_entry_files.py 16 > gpurun_out/r3/entry_files2.log 2>&1; tail -n 1 gpurun_out/r3/entry_files2.log | cut -c1-900
