for args in "--model xlm-roberta --batch 32" "--model xlm-roberta --batch 256" "--batch 64 --seq_len 256 --n_best 10 --add_l2_loss" "--model xlm-roberta-large --batch 64 --seq_len 256" "--model xlm-roberta-large --batch 64 --seq_len 256 --dtype fp8w"; do
  echo "== $args: $(python bench.py $args --no_cpu_baseline --no_roofline --steps 10 --warmup 3 2>&1 | grep 'timed region')"
done
