cd $GRAFT_REPO_ROOT
timeout 1200 bash scripts/collect_profiles.sh r3v1 > gpurun_out/r3r_collect.log 2>&1
timeout 900 bash scripts/collect_profiles_1e6.sh r3_1e6 > gpurun_out/r3r_collect_1e6.log 2>&1
timeout 600 bash scripts/prof_c4.sh r3_c4 > gpurun_out/r3r_c4.log 2>&1
tail -3 gpurun_out/r3r_collect.log; ls gpurun_out/r3v1 gpurun_out/r3_1e6 | head -40; du -sh gpurun_out/r3v1 gpurun_out/r3_1e6 gpurun_out/r3_c4
