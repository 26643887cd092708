#!/bin/bash
# everything the round's profile files come from, at the round's last code, one box
cd $GRAFT_REPO_ROOT
bash scratch/profile_round.sh r04 > gpurun_out/profile_round.log 2>&1 || { tail -20 gpurun_out/profile_round.log; exit 1; }
tail -30 gpurun_out/profile_round.log
python scratch/filters_bench.py > gpurun_out/r04_other_filters.json 2> gpurun_out/filters_bench.err
python scratch/chain_bench.py > gpurun_out/r04_chain_config5.json 2> gpurun_out/chain_bench.err
python scratch/sor_small.py > gpurun_out/r04_sor_small_clouds.txt 2>&1
python scratch/partition_probe.py > gpurun_out/r04_permuted_partition.txt 2>&1
ls -la gpurun_out/ | tail -12
