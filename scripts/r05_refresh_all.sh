# round-5 refresh: GPU suite + headline profiles (refresh_profiles.sh) + the round-5 additions (refresh_profiles_r05.sh)
RND=r05 bash scripts/refresh_profiles.sh > gpurun_out/refresh_r05.log 2>&1
bash scripts/refresh_profiles_r05.sh > gpurun_out/refresh5_r05.log 2>&1
