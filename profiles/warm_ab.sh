for e in "ACF_HIP_LEVEL_WARM=32 ACF_HIP_SMOOTH_WARM=48" "ACF_HIP_LEVEL_WARM=64 ACF_HIP_SMOOTH_WARM=64" "ACF_HIP_LEVEL_WARM=96 ACF_HIP_SMOOTH_WARM=96" "ACF_HIP_LEVEL_WARM=96 ACF_HIP_SMOOTH_WARM=48"; do echo "== $e"; env $e python profiles/latency_breakdown.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['segments_auto']; print(round(d['latency_ms'],3), d['us_per_launch_in_order']['k_smooth_vec'], d['us_per_launch_in_order']['k_level(fused)'])"; done
KERNELS="k_smooth_vec" bash profiles/ab.sh "ACF_HIP_SMOOTH_WARM=48" "ACF_HIP_SMOOTH_WARM=96" "ACF_HIP_SMOOTH_WARM=64"
