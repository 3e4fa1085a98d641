# A/B of the round-4 training-step changes (one gpurun call): CNNRNNModelLarge step, B = 16
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
CMD="bench.py --mode train --model cnn_rnn_large --batch 16 --steps 8 --warmup 2"
for tag in all_on ws0 direct0 splitk0 all_off; do
  case $tag in
    all_on) E="MT_X=1";;
    ws0) E="MT_TRAIN_WS_CACHE=0";;
    direct0) E="MT_DIRECT_GRADS=0";;
    splitk0) E="MT_GEMM_SPLITK=0";;
    all_off) E="MT_TRAIN_WS_CACHE=0 MT_DIRECT_GRADS=0 MT_GEMM_SPLITK=0";;
  esac
  echo "== $tag ($E)" >> $O/ab.txt
  env $E timeout -k 10 120 python3 $CMD > $O/$tag.json 2> $O/$tag.err
  grep -o '"ms_per_step": [0-9.]*' $O/$tag.json >> $O/ab.txt
  grep "host enqueue" $O/$tag.err >> $O/ab.txt
done
cat $O/ab.txt
