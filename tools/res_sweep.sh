run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --resident 2 --sustain 0 > gpurun_out/res_$tag.json 2>gpurun_out/res_$tag.err; }
run base A=1
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run sdma HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0
run base2 A=1
python3 - <<'PY'
import json
for r in ("base","blit0","sdma","base2"):
    try:
        d=json.loads(open(f"gpurun_out/res_{r}.json").read().strip().splitlines()[-1])
        print(r, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels_ms"])
    except Exception as e:
        print(r, "failed", e, open(f"gpurun_out/res_{r}.err").read()[-300:])
PY
