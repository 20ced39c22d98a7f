#!/bin/bash
# sample the GPU clock / power while the benchmark runs (is the chip at its power limit during the GEMM-heavy step?)
mkdir -p gpurun_out/clk
python bench.py --steps 1500 --warmup 5 --no_cpu_baseline > gpurun_out/clk/bench.json 2> gpurun_out/clk/bench.err &
BP=$!
sleep 22
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
  echo ---
  sleep 0.7
done
wait $BP
cut -c1-160 gpurun_out/clk/bench.json
echo idle:
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
