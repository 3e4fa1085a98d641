cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python tools/lstm_diag.py 128 938 512 > gpurun_out/lstm_diag.txt 2>&1; grep -v amdgpu gpurun_out/lstm_diag.txt | tail -14 | cut -c1-200
