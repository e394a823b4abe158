#!/bin/bash
# Regenerates the committed round-4 measurement artefacts on the GPU box (run through gpurun from the repo root) into
# gpurun_out/final4/; copy what is to be judged into profiles/ as r4_final_<name> (profiles/profile_index.json names the files
# bench.py reads for `in_replay_us` and `traffic`).
export TMPDIR=/tmp
O=gpurun_out/final4
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.log
# the bench command under the profiler (bench.py's in_replay_us comes from this summary)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --no-cpu-baseline --no-batch32 > $O/bench_under_rocprof.log 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 profiles/summarize_trace.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) > $O/forward_breakdown.txt
python3 profiles/iteration_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 100 > $O/iteration_timeline.txt
python3 profiles/encoder_timeline.py $(find $O/prof -name "t_kernel_trace.csv" | head -1) 100 > $O/encoder_timeline.txt
rm -rf $O/prof
# PMC passes: counters in their own runs, never together with a trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-batch32 > /dev/null 2>&1
done
python3 profiles/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-batch32 > /dev/null 2>&1
python3 profiles/mfma_busy.py $O/pmc_mfma > $O/mfma_busy.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
python bench.py 2> $O/bench_stderr.log | tail -1 > $O/bench_n1.json
python bench.py --batch 32 --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_batch32.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof32 -o t -- python3 bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find $O/prof32 -name "t_kernel_stats.csv" | head -1) $O/kernel_stats_batch32.csv
rm -rf $O/prof32
python profiles/time_sizes.py 512x1024:12 640x1280:32 480x960:12 256x512:4 > $O/time_sizes.txt 2>&1
# training step: the captured graph, the eager loop node, the per-node tape of round 3, batch 8
python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 > $O/train_step_time.json
python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 > $O/train_step_time_eager.json
PRIORFLOW_TRAIN_LOOP=0 PRIORFLOW_TRAIN_FORK=0 python profiles/time_train_step.py --steps 10 2>/dev/null | tail -1 > $O/train_step_time_tape.json
python profiles/time_train_step.py --steps 5 --batch 8 --graph 2>/dev/null | tail -1 > $O/train_step_time_batch8.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/proft -o t -- python3 profiles/time_train_step.py > $O/train_trace.log 2>&1
cp $(find $O/proft -name "t_kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv
python3 profiles/train_sequence.py $(find $O/proft -name "t_kernel_trace.csv" | head -1) > $O/train_sequence.txt     # launches / PyTorch kernels of one eager step
rm -rf $O/proft
rocprofv3 --kernel-trace --output-format csv -d $O/proftg -o t -- python3 profiles/time_train_step.py --steps 3 --warmup 1 --graph > /dev/null 2>&1
python3 profiles/train_phases.py $(find $O/proftg -name "t_kernel_trace.csv" | head -1) > $O/train_phases.txt           # one replayed step by phase
rm -rf $O/proftg
for i in 1 2; do for p in 1 0; do
  PRIORFLOW_GRAD_SINK=$p PRIORFLOW_TRAIN_BN_FUSED=$p python profiles/time_train_step.py --steps 10 --graph 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('graphed step, grad sink + fused BN = $p:', d['ms_per_step'], 'ms')"
done; done > $O/ab_train_sink.txt
PRIORFLOW_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_gloo2.json
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tests/run_train_2rank.py 2>&1 | grep -E "rank|checksums" > $O/train_2rank_check.txt
# same-box A/Bs, interleaved
for i in 1 2; do for p in 0 1; do
  PRIORFLOW_STEM_DIRECT=$p python bench.py --no-cpu-baseline --no-batch32 --steps 60 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1  stem_direct=$p', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
  PRIORFLOW_STEM_DIRECT=$p python bench.py --batch 32 --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=32 stem_direct=$p', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done; done > $O/ab_stem.txt
for i in 1 2; do for p in 0 1; do
  PRIORFLOW_LOOKUP_WIN=$p python bench.py --batch 32 --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=32 lookup_win=$p', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done; done > $O/ab_lookup_win_batch32.txt
cat $O/pytest_gpu.log; cut -c1-300 $O/bench_n1.json; head -3 $O/forward_breakdown.txt | cut -c1-200; cat $O/time_sizes.txt; cut -c1-200 $O/bench_batch32.json; cat $O/ab_stem.txt; cut -c1-250 $O/train_step_time*.json; head -1 $O/train_sequence.txt; cat $O/ab_train_sink.txt
