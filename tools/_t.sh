cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for c in "16 937 512" "16 937 256" "4 300 512" "32 400 512" "24 400 512" "40 300 320" "9 200 176"; do
  timeout -k 10 120 python tools/bptt_ab.py $c 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
