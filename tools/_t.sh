cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pt.txt 2>&1; tail -3 gpurun_out/pt.txt | cut -c1-300
timeout -k 10 300 python bench.py --mode train --batch 16 --steps 20 --warmup 5 > gpurun_out/train_small.txt 2>&1; tail -1 gpurun_out/train_small.txt | cut -c1-230
