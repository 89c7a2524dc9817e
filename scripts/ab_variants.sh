#!/bin/bash
# Developer tool: the default library against A/B builds of tools/build_variant.sh on ONE box:  scripts/ab_variants.sh <outdir> tag...
out=$1; shift
mkdir -p $out
one() { tag=$1; lib=$2
  env QD_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps 120 --warmup 24 --no-cpu-baseline --no-ecology-leg --profile-kernel ocean_tail > $out/$tag.json 2>$out/$tag.err || { echo "$tag FAILED"; return; }
  python - <<PY
import json
j=json.loads(open("$out/$tag.json").read().strip().splitlines()[-1]); print("$tag", round(j["ms_per_step"],4), j["config"]["ocean_n_sub"], "tail us", round(j["roofline"]["avg_kernel_ms"]*1e3,2), "k_ocn_stream us", round(j["roofline_ocean_substep"]["avg_kernel_ms"]*1e3,2))
PY
}
one default qingdai_amd/libqingdai_hip.so
for t in "$@"; do one $t tools/variants/libqingdai_hip_$t.so; done
one default_again qingdai_amd/libqingdai_hip.so
