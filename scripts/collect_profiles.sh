#!/bin/bash
# Collect the measured evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r02
# Writes under gpurun_out/<tag>_*; scripts/refresh_profiles.py <tag> condenses them into profiles/.
tag=${1:-r03}
part=${2:-all}      # a: headline bench + rocprofv3 passes; b: other configurations; c: size sweep, whole fits (gpurun calls are capped at 20 min)
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
set -x
if [ "$part" = "all" ] || [ "$part" = "a" ]; then
python bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err
python bench.py --dtype f32 --no-cpu-baseline > $out/${tag}_bench_f32.json 2>/dev/null
python bench.py --dtype mixed --no-cpu-baseline > $out/${tag}_bench_mixed.json 2>/dev/null
rm -rf $out/${tag}_kt $out/${tag}_pmcf $out/${tag}_pmcw $out/${tag}_pmcm
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-in-flight > $out/${tag}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmcf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-in-flight > $out/${tag}_pmcf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmcw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-in-flight > $out/${tag}_pmcw.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmcm -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-in-flight > $out/${tag}_pmcm.log 2>&1
# keep only what refresh_profiles.py reads (the raw traces are large)
find $out/${tag}_kt -name "*kernel_trace.csv" -delete
# the truncated-rank closure (the regime of the reference's default tolerance): per-kernel rows of its own
rm -rf $out/${tag}_kt_trunc
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_trunc -- python bench.py --config trunc --steps 10 --warmup 2 --no-cpu-baseline > $out/${tag}_kt_trunc.log 2>&1
find $out/${tag}_kt_trunc -name "*kernel_trace.csv" -delete
fi
if [ "$part" = "all" ] || [ "$part" = "b" ]; then
: > $out/${tag}_configs.jsonl
python bench.py --config n4096 --steps 20 2>/dev/null >> $out/${tag}_configs.jsonl
python bench.py --config cells64 --steps 3 --warmup 1 2>/dev/null >> $out/${tag}_configs.jsonl
python bench.py --config thetagrid --steps 1 --warmup 0 2>/dev/null >> $out/${tag}_configs.jsonl
python bench.py --config thetagrid --dtype f32 --steps 1 --warmup 0 2>/dev/null >> $out/${tag}_configs.jsonl
python bench.py --config thetagrid --dtype f64 --steps 1 --warmup 0 2>/dev/null >> $out/${tag}_configs.jsonl
python bench.py --config trunc --steps 20 --warmup 3 2>/dev/null >> $out/${tag}_configs.jsonl
python bench.py --config sparse --steps 20 --warmup 3 2>/dev/null >> $out/${tag}_configs.jsonl
fi
if [ "$part" = "all" ] || [ "$part" = "c" ]; then
: > $out/${tag}_size_sweep.jsonl
for n in 1024 2048 4096 6144 8192 12288 16384 32768; do python bench.py --n $n --steps 6 --warmup 2 --no-cpu-baseline --no-in-flight 2>/dev/null >> $out/${tag}_size_sweep.jsonl; done
: > $out/${tag}_whole_fits.log
python examples/one_cell_fit.py --n 512 --d 64 >> $out/${tag}_whole_fits.log 2>&1
python examples/one_cell_fit.py --n 1024 --d 256 >> $out/${tag}_whole_fits.log 2>&1
python examples/one_cell_fit.py --n 2048 --d 256 >> $out/${tag}_whole_fits.log 2>&1
python examples/one_cell_fit.py --n 2500 --ntilde 1536 --d 256 --maxiter 10 --nestep 5 --nmstep 6 >> $out/${tag}_whole_fits.log 2>&1
python examples/one_cell_fit.py --n 4096 --d 256 >> $out/${tag}_whole_fits.log 2>&1
python examples/one_cell_fit.py --n 8192 --d 256 >> $out/${tag}_whole_fits.log 2>&1
python examples/one_cell_fit.py --n 8192 --d 256 --tol 1e-14 >> $out/${tag}_whole_fits.log 2>&1
GPFIT_FORCE_EIGH=1 python examples/one_cell_fit.py --n 8192 --d 256 --tol 1e-14 >> $out/${tag}_whole_fits.log 2>&1
python scripts/scratch/dev_projected.py 4096 >> $out/${tag}_whole_fits.log 2>&1
python scripts/whole_fit_breakdown.py 8192 256 $out/${tag}_whole_fit_breakdown.json > /dev/null 2>&1
python examples/active_learning.py --pool 200 --start 40 --iterations 6 > $out/${tag}_active_loop.log 2>&1
python examples/active_learning.py --pool 200 --start 40 --iterations 6 --notebook-step >> $out/${tag}_active_loop.log 2>&1
fi
tail -3 $out/${tag}_bench_full.err
