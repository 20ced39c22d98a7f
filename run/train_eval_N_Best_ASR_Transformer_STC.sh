#!/bin/bash
# Fine-tune / evaluate the N-best ASR transformer STC model on MI355X (HIP path).
# Same knobs as the reference launcher (run/train_eval_N_Best_ASR_Transformer_STC.sh there); one process per GPU:
#   NGPU=8 bash run/train_eval_N_Best_ASR_Transformer_STC.sh
set -e
dataset=${dataset:-dstc2}
dataroot=${dataroot:-dstc2_data/processed_data/raw}
exp_path=${exp_path:-exp/exp_bert_sep_segment_ids/}
pre_trained_model=${pre_trained_model:-bert}      # bert | xlm-roberta
device=${device:-0}
dp=${dp:-0.3}; bert_dp=${bert_dp:-0.1}
bs=${bs:-16}; me=${me:-50}; mn=${mn:-5.0}
lr=${lr:-3e-5}; bert_lr=${bert_lr:-3e-5}; wp=${wp:-0.1}
seed=${seed:-999}
coverage=${coverage:-1.0}                          # (0,1]: stratified share of the training split
NGPU=${NGPU:-1}
extra="$@"                                         # e.g. --add_l2_loss --without_system_act --init_checkpoint bert.pt --vocab vocab.txt

args="--dataset ${dataset} --dataroot ${dataroot} --deviceId ${device} --random_seed ${seed} --l2 1e-8 --dropout ${dp} \
  --bert_dropout ${bert_dp} --optim_choice bertadam --lr ${lr} --bert_lr ${bert_lr} --warmup_proportion ${wp} \
  --init_type uf --init_range 0.02 --batchSize ${bs} --max_norm ${mn} --max_epoch ${me} --experiment ${exp_path} \
  --pre_trained_model ${pre_trained_model} --coverage ${coverage} --add_segment_ids ${extra}"
if [ "${NGPU}" -gt 1 ]; then
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node ${NGPU} --master-addr 127.0.0.1 --master-port ${PORT:-29500} n_best_asr_bert.py ${args}
else
  python3 n_best_asr_bert.py ${args}
fi
