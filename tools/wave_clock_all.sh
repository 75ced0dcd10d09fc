mkdir -p gpurun_out/r2q
for k in 3 4 41 42 5; do DQL_LIB_PATH=$PWD/dql_multirotor_landing_amd/csrc/libdql_hip_clock$k.so python tools/exp_wave_clock.py 4096 | head -1 > gpurun_out/r2q/clock$k.jsonl; done
