#!/bin/bash
# One parametrised runner for every GPU call of a round (replaces the per-call r0N_callM.sh scripts of rounds 2-4):
#
#     gpurun --timeout 900 -- 'bash tools/lab/run.sh <label> <step> [<step> ...]'
#
# Every step writes <step>.log (+ its JSON) under gpurun_out/<label>/ and prints a short tail; parameters travel in the environment
# (the step's own tool documents them).  Steps:
#   gemm_ab      tools/gemm_sched_ab.py        SCHEDS=8273,24657 SHAPES=qkv,proj,fc1,fc2 ROUNDS=6 B=256
#   tok_ab       tools/tok_ab.py               TOK_SETS="|gemm_sched=8273" (|-separated option sets) ROUNDS=5
#   prefill_ab   tools/prefill_bench.py        AB="gemm_sched=8273|gemm_sched=-1" MODEL=14b|8b
#   decode_ab    tools/decode_ab.py            DECODE_SETS=...
#   attn         tools/attn_bench.py
#   pytest       python -m pytest $PYTEST_ARGS (default: the whole -m gpu suite); PYTEST_K="a or b" adds -k
#   bench        python bench.py $BENCH_ARGS  -> bench.json
#   prof_bench   rocprofv3 --kernel-trace --stats of bench.py $BENCH_ARGS
#   smoke        __graft_entry__.smoke()
#   py:<file>    python <file> (any tool under tools/, arguments in PY_ARGS)
#   sh:<file>    bash <file> (tools/pmc_qkv.sh, tools/pmc_skinny.sh ...)
#   prof:<file>  the same under rocprofv3 --kernel-trace --stats -> <tool>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
export TMPDIR=/tmp
label=$1; shift
O=gpurun_out/$label
mkdir -p "$O"
show() { grep -v "^/opt\|amdgpu.ids" "$1" | cut -c1-400 | tail -n "${2:-8}"; }
for step in "$@"; do
    echo "=== $step ($(date +%T))"
    case $step in
        gemm_ab)
            SCHEDS=${SCHEDS:-8273,24657} SHAPES=${SHAPES:-qkv,proj,fc1,fc2} ROUNDS=${ROUNDS:-6} OUT=$O/gemm_sched_ab.json \
                timeout 600 python tools/gemm_sched_ab.py > $O/gemm_ab.log 2>&1; echo "rc=$?" >> $O/gemm_ab.log
            show $O/gemm_ab.log 8 ;;
        tok_ab)
            IFS='|' read -r -a sets <<< "${TOK_SETS:-|gemm_sched=8273}"
            OUT=$O/tok_ab.json ROUNDS=${ROUNDS:-5} timeout 900 python tools/tok_ab.py "${sets[@]}" > $O/tok_ab.log 2>&1; echo "rc=$?" >> $O/tok_ab.log
            show $O/tok_ab.log 30 ;;
        prefill_ab)
            AB=${AB:-"gemm_sched=8273|gemm_sched=-1"} ROUNDS=${ROUNDS:-5} timeout 600 python tools/prefill_bench.py > $O/prefill_ab_${MODEL:-14b}.log 2>&1
            echo "rc=$?" >> $O/prefill_ab_${MODEL:-14b}.log; show $O/prefill_ab_${MODEL:-14b}.log 6 ;;
        decode_ab)
            timeout 600 python tools/decode_ab.py ${DECODE_SETS} > $O/decode_ab.log 2>&1; echo "rc=$?" >> $O/decode_ab.log; show $O/decode_ab.log 12 ;;
        attn)
            timeout 300 python tools/attn_bench.py > $O/attn.log 2>&1; echo "rc=$?" >> $O/attn.log; show $O/attn.log 12 ;;
        pytest)
            if [ -n "$PYTEST_K" ]; then timeout ${PYTEST_TIMEOUT:-1500} python -m pytest ${PYTEST_ARGS:-tests -m gpu -q -x} -k "$PYTEST_K" > $O/pytest.log 2>&1
            else timeout ${PYTEST_TIMEOUT:-1500} python -m pytest ${PYTEST_ARGS:-tests -m gpu -q -x} > $O/pytest.log 2>&1; fi; echo "pytest rc=$?" >> $O/pytest.log
            show $O/pytest.log 12 ;;
        bench)
            timeout 600 python bench.py ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
            tail -c 3000 $O/bench.json; show $O/bench.err 3 ;;
        prof_bench)
            (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py ${BENCH_ARGS:---steps 5 --warmup 2} > $R/$O/prof_bench.log 2>&1)
            echo "rc=$?" >> $O/prof_bench.log; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 ;;
        smoke)
            timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; show $O/smoke.log 4 ;;
        prof:*)                                         # rocprofv3 --kernel-trace --stats of any tool: prof:tools/tok_trace.py (arguments in PY_ARGS)
            f=${step#prof:}; n=$(basename "$f" .py)
            (cd /tmp && timeout ${PY_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$n -o $n -- python $R/$f ${PY_ARGS} > $R/$O/prof_$n.log 2>&1)
            echo "rc=$?" >> $O/prof_$n.log
            st=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" $O/${n}_kernel_stats.csv && head -${PY_TAIL:-16} "$st" | cut -c1-200
            find $O/prof_$n -name "*.db" -delete 2>/dev/null; find $O/prof_$n -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null ;;
        sh:*)                                           # any shell tool: sh:tools/pmc_qkv.sh
            f=${step#sh:}; n=$(basename "$f" .sh)
            timeout ${PY_TIMEOUT:-900} bash "$f" ${PY_ARGS} > $O/$n.log 2>&1; echo "rc=$?" >> $O/$n.log; show $O/$n.log ${PY_TAIL:-24} ;;
        py:*)
            f=${step#py:}; n=$(basename "$f" .py)
            timeout ${PY_TIMEOUT:-600} python "$f" ${PY_ARGS} > $O/$n.log 2>&1; echo "rc=$?" >> $O/$n.log; show $O/$n.log ${PY_TAIL:-20} ;;
        *) echo "unknown step $step" ;;
    esac
done
echo "=== done ($(date +%T))"
