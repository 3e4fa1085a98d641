cd "$(dirname "$0")/.."
timeout -k 10 120 python tools/bptt_ab.py 16 937 512 && timeout -k 10 120 python tools/bptt_ab.py 16 937 256 && timeout -k 10 120 python tools/bptt_ab.py 4 300 512 && timeout -k 10 120 python tools/bptt_ab.py 32 400 512 && timeout -k 10 120 python tools/bptt_ab.py 40 300 320 && timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -5
