# GPU session r06/35: round profile of the tree so far (bench line, kernel trace, PMC traffic, SQ counters, step_profile.json) -- tag r06y
bash tools/run_prof.sh r06y > gpurun_out/r06y_run_prof.log 2>&1; tail -40 gpurun_out/r06y/step_profile.log; head -12 gpurun_out/r06y/r06y_pmc_traffic.md; cut -c1-600 gpurun_out/r06y/bench.json
