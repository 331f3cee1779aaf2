# the round's closing job: full GPU suite, profiles (rocprofv3 stats, steady state, PMC, PVT workloads), the driver's default bench
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/t_final.log 2>&1; echo rc=$? >> gpurun_out/t_final.log; tail -3 gpurun_out/t_final.log
bash tools/job_profiles.sh
bash tools/job_driver_like.sh
