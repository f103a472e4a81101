# round-5 refresh of the BASELINE configs nobody timed since round 1: 224^2 finetune, config 2 (224^2 two-view
# pretrain bs 32) in BOTH storage builds, config 4, and the default workload in the bf16 build.  usage: bash tools/bench_configs_r5.sh <outdir>
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r5cfg}
mkdir -p $O
cd $R
python3 bench.py --res 224 --no-decode --steps 20 --warmup 5 > $O/ft224_f16.json 2> $O/ft224_f16.err && \
python3 bench.py --workload pretrain --res 224 --batch 32 --steps 20 --warmup 5 > $O/pt224_f16.json 2> $O/pt224_f16.err && \
EVK_STORE=bf16 python3 bench.py --workload pretrain --res 224 --batch 32 --steps 20 --warmup 5 > $O/pt224_bf16.json 2> $O/pt224_bf16.err && \
python3 bench.py --config 4 --steps 20 --warmup 5 > $O/cfg4_f16.json 2> $O/cfg4_f16.err && \
EVK_STORE=bf16 python3 bench.py --steps 20 --warmup 5 --no-decode > $O/ft384_bf16.json 2> $O/ft384_bf16.err && \
EVK_STORE=bf16 python3 bench.py --res 224 --no-decode --steps 20 --warmup 5 > $O/ft224_bf16.json 2> $O/ft224_bf16.err
echo rc=$?
for f in $O/*.json; do echo $f; cut -c1-400 $f; done
