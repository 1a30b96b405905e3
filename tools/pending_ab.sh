#!/bin/bash
# EXECUTED in round 4, session 1 (logs: profiles/r04/ab/{train,hrnet,config5,pytest}_*.log); RIH_WGRAD_GROUP_T128 and RIH_BN_LASTBLOCK
# were removed after it (measured neutral / slower) -- kept as the record of what produced those logs, not runnable any more.
# The experiments that were BUILT but not measured when round 3 ran out of GPU minutes, as one GPU-box session (~12 min):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pending_ab.sh'
# 1. RIH_SKIP_DEAD_MID=1 -- the finest mid convolution, whose output decoder.forward drops: the gated GPU
#    parity test, then the training step and the fp16 inference step (--config5) with and without it.
# 2. RIH_WGRAD_GROUP_T128=128 / 256 -- 128x128 tiles for the large grouped weight gradients (item 3b).
# 3. RIH_BN_LASTBLOCK=1 -- BatchNorm reductions without their finishing launch (item 3c): parity tests with the flag set, then
#    the ResNet50 and HRNet-W32 steps with and without it.
# 3'. RIH_GEMM_DROPOUT=1 -- dropout behind the decoder's Linears in the GEMM epilogue (item 3d).
# Each bench line is the last line of its log under gpurun_out/pending/.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/pending
mkdir -p "$OUT"
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run() { name=$1; shift; echo "== $name: $*"; ( timeout 600 "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-260; }
run pytest_dead_mid env RIH_SKIP_DEAD_MID=1 python -m pytest tests -q -m gpu -x -k "dead_mid or model_eval_matches or model_train_matches or fp16_backbone"
run train_base python bench.py $Q
run train_skip_dead_mid env RIH_SKIP_DEAD_MID=1 python bench.py $Q
run train_t128 env RIH_WGRAD_GROUP_T128=128 python bench.py $Q
run train_t256 env RIH_WGRAD_GROUP_T128=256 python bench.py $Q
run pytest_bn_lastblock env RIH_BN_LASTBLOCK=1 python -m pytest tests -q -m gpu -x -k "batchnorm or conv_bn or model_train_matches or hrnet_eval or side_streams"
run train_bn_lastblock env RIH_BN_LASTBLOCK=1 python bench.py $Q
run pytest_gemm_dropout env RIH_GEMM_DROPOUT=1 python -m pytest tests -q -m gpu -x -k "linear_dropout or dropout or model_train_matches or hipgraph"
run train_gemm_dropout env RIH_GEMM_DROPOUT=1 python bench.py $Q
run train_all_four env RIH_GEMM_DROPOUT=1 RIH_BN_LASTBLOCK=1 RIH_SKIP_DEAD_MID=1 RIH_WGRAD_GROUP_T128=128 python bench.py $Q
run train_base2 python bench.py $Q
run hrnet_base python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run hrnet_bn_lastblock env RIH_BN_LASTBLOCK=1 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run config5_base python bench.py --config5
run config5_skip_dead_mid env RIH_SKIP_DEAD_MID=1 python bench.py --config5
# 4. where the HRNet-W32 step spends its time now that its branches overlap (no kernel trace of it exists for round 3)
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_hr && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hr -o step -- \
    python $R/bench.py --encoder hrnet32 --steps 5 --warmup 2 --no-cpu-baseline --no-reference-loop --no-roofline ) > "$OUT/prof_hrnet.log" 2>&1
cp /tmp/prof_hr/step_kernel_stats.csv "$OUT/hrnet_kernel_stats.csv" 2>/dev/null
cp /tmp/prof_hr/step_kernel_trace.csv "$OUT/hrnet_kernel_trace.csv" 2>/dev/null
python tools/step_from_trace.py "$OUT/hrnet_kernel_trace.csv" --mark nchw_to_nhwc_kernel > "$OUT/step_trace_hrnet.txt" 2>&1
rm -f "$OUT/hrnet_kernel_trace.csv"      # tens of MB: the summary above is what is kept
echo done
