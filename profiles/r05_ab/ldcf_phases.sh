# k_ldcf_tile with a phase compiled out (profiles/build_variant.sh NAME -DACF_LDCF_NO_FILL / _NO_CONV / _NO_PASSES): solo ms per 16 4K frames
run() { python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-verify --no-repeats 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline']['solo']['kernels_ms_per_batch']['k_ldcf_tile'])"; }
for xo in "" "ACF_HIP_LDCF_XO8=1"; do
for v in "" _nofill _noconv _nopass; do
  echo "== $xo lib$v"; env $xo ACF_HIP_LIB=acf_amd/libacf_hip$v.so python -c "print()" >/dev/null
  if [ -z "$v" ]; then env $xo bash -c "$(declare -f run); run"; else env $xo ACF_HIP_LIB=acf_amd/libacf_hip$v.so bash -c "$(declare -f run); run"; fi
done; done
