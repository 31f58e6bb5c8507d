#!/bin/bash
# tools/training_consoles.sh <tag>: end-to-end training runs through `python -m pime_amd.train` (the reference's CLI), consoles into
# gpurun_out/<tag>_*_training_console.txt: the headline pH config to 4e8 env-steps, the reference's two water-tank script blocks
# (ResidualPPO / ResidualIntegratorModularPPO, net_dim 256) to 1e8, and residual TD3 on 4096 tank lanes.
# tools/training_consoles.sh <tag> td3: the TD3 run only
TAG=$1; ONLY=$2; OUT=gpurun_out; mkdir -p $OUT; cd ${GRAFT_REPO_ROOT:-/root/repo}
LOG=/tmp/pime_logs
run() {
  name=$1; shift
  if [ "$ONLY" = td3 ] && [ "$name" != wt_residual_td3 ]; then return; fi
  t0=$(date +%s)
  timeout -k 10 400 python -m pime_amd.train "$@" --log_root $LOG > $OUT/${TAG}_${name}_training_console.txt 2>&1
  echo "wall clock of the whole command (imports, table, graph capture, evaluations included): $(( $(date +%s) - t0 )) s" >> $OUT/${TAG}_${name}_training_console.txt
  tail -4 $OUT/${TAG}_${name}_training_console.txt
}
run ph --algo ResidualIntegratorModularPPO --fix_K --env PH1DChangingParamUniformGoalIntegrator-SqaureDistance-v35 --net_dim 128 \
    --num_envs 16384 --target_step 819200 --batch_size 65536 --repeat_times 8 --lambda_gae_adv 0.99 --break_step 400000000 \
    --eval_times1 4096 --eval_times2 4096
run wt_stacking10_width256 --algo ResidualPPO --fix_K --env NonLinearWaterTankChangingParamUniformGoalStacking10-SquareDistance-v2 \
    --reward_type distance --net_dim 256 --num_envs 4096 --target_step 819200 --batch_size 65536 --repeat_times 8 --break_step 100000000 \
    --eval_times1 4096 --eval_times2 4096
run wt_integrator_modular_width256 --algo ResidualIntegratorModularPPO --fix_K \
    --env NonLinearWaterTankChangingParamUniformGoalIntegrator-SquareDistance-v2 --reward_type distance --net_dim 256 --num_envs 4096 \
    --target_step 819200 --batch_size 65536 --repeat_times 8 --break_step 100000000 --eval_times1 4096 --eval_times2 4096
run wt_residual_td3 --algo ResidualTD3 --fix_K --env NonLinearWaterTankChangingParamUniformGoalIntegrator-SquareDistance-v2 \
    --reward_type distance --net_dim 128 --num_envs 4096 --target_step 819200 --batch_size 4096 --break_step 20000000 \
    --eval_times1 4096 --eval_times2 4096
