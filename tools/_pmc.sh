mkdir -p gpurun_out/r10
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --no_cpu_baseline 2>/dev/null | tee gpurun_out/r10/bench.json | cut -c1-200
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r10/fetch -- python tools/wgrad_once.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r10/write -- python tools/wgrad_once.py 3 > /dev/null 2>&1
