#!/bin/bash
mkdir -p gpurun_out/r5
C="--steps 12 --warmup 3 --no-cpu-baseline --no-alt-dtype --no-attainable --no-roofline --model-type vitgan --batch 32"
run() { name=$1; shift; env "$@" python bench.py $C > gpurun_out/r5/c3_$name.json 2> gpurun_out/r5/c3_$name.err; tail -1 gpurun_out/r5/c3_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['step_ms_main_stream'])"; }
run off FFVC_SMALLM=0
run on FFVC_SMALLM=1
run nt_only FFVC_SMALLM=1 FFVC_SMALLM_TT=0
run tt_only FFVC_SMALLM=0 FFVC_SMALLM_TT=1
run on_nosplit FFVC_SMALLM=1 FFVC_SK_FIXUP=0
C="$C --no-side-stream"; run on_noside FFVC_SMALLM=1; C="${C% --no-side-stream}"
run on_again FFVC_SMALLM=1
