// acf_hip_loader.h — thin dlopen() loader of libacf_hip.so (include/acf_hip.h).
//
// The host side of the backend is plain C++ (g++; no HIP headers, no torch):
// every device action goes through the C ABI resolved here at run time, the
// way the reference's GL backend is reached through ogles_gpgpu behind
// acf::GLDetector (src/app/acf/GLDetector.cpp:117-130).  There is no CPU
// fallback: if the library or a symbol is missing, load() throws.
#pragma once

#include "../../include/acf_hip.h"

#include <string>

namespace acf
{
namespace hip
{

struct Api
{
    void* handle = nullptr;
#define ACF_HIP_FN(name) decltype(&::name) name = nullptr;
    ACF_HIP_FN(acf_hip_create)
    ACF_HIP_FN(acf_hip_device_count)
    ACF_HIP_FN(acf_hip_destroy)
    ACF_HIP_FN(acf_hip_abi_version)
    ACF_HIP_FN(acf_hip_last_error)
    ACF_HIP_FN(acf_hip_set_option)
    ACF_HIP_FN(acf_hip_get_scales)
    ACF_HIP_FN(acf_hip_plan_levels)
    ACF_HIP_FN(acf_hip_set_model)
    ACF_HIP_FN(acf_hip_plan)
    ACF_HIP_FN(acf_hip_num_levels)
    ACF_HIP_FN(acf_hip_get_levels)
    ACF_HIP_FN(acf_hip_get_ldcf_levels)
    ACF_HIP_FN(acf_hip_pyramid_floats)
    ACF_HIP_FN(acf_hip_get_lambdas)
    ACF_HIP_FN(acf_hip_pyramid)
    ACF_HIP_FN(acf_hip_detect)
    ACF_HIP_FN(acf_hip_run)
    ACF_HIP_FN(acf_hip_run_host)
    ACF_HIP_FN(acf_hip_pyramid_u8)
    ACF_HIP_FN(acf_hip_run_u8)
    ACF_HIP_FN(acf_hip_resize_dims)
    ACF_HIP_FN(acf_hip_set_input_resize)
    ACF_HIP_FN(acf_hip_op_resize_u8)
    ACF_HIP_FN(acf_hip_stream_open)
    ACF_HIP_FN(acf_hip_stream_submit)
    ACF_HIP_FN(acf_hip_stream_collect)
    ACF_HIP_FN(acf_hip_stream_close)
    ACF_HIP_FN(acf_hip_host_alloc)
    ACF_HIP_FN(acf_hip_host_free)
    ACF_HIP_FN(acf_hip_set_nms)
    ACF_HIP_FN(acf_hip_op_nms)
    ACF_HIP_FN(acf_hip_get_detections)
    ACF_HIP_FN(acf_hip_get_hits)
    ACF_HIP_FN(acf_hip_get_raw_detections)
    ACF_HIP_FN(acf_hip_export_detections)
    ACF_HIP_FN(acf_hip_synchronize)
    ACF_HIP_FN(acf_hip_get_repairs)
    ACF_HIP_FN(acf_hip_profile_get)
    ACF_HIP_FN(acf_hip_read_level)
    ACF_HIP_FN(acf_hip_read_rank_level)
    ACF_HIP_FN(acf_hip_rank_cells_host)
    ACF_HIP_FN(acf_hip_read_tap)
    ACF_HIP_FN(acf_hip_op_rgb_convert)
    ACF_HIP_FN(acf_hip_op_conv_tri)
    ACF_HIP_FN(acf_hip_op_gradient_mag)
    ACF_HIP_FN(acf_hip_op_gradient_hist)
    ACF_HIP_FN(acf_hip_chns_compute)
    ACF_HIP_FN(acf_hip_set_x86_tables)
    ACF_HIP_FN(acf_hip_selftest_x86)
    ACF_HIP_FN(acf_hip_op_im_resample)
    ACF_HIP_FN(acf_hip_op_acf_detect1)
    ACF_HIP_FN(acf_hip_op_acf_detect1_u8)
    ACF_HIP_FN(acf_hip_thrs_u8)
    ACF_HIP_FN(acf_hip_op_evaluate)
#undef ACF_HIP_FN
};

// Loads (once per process) and returns the API table.  `path` empty: the
// ACF_HIP_LIBRARY environment variable, else "libacf_hip.so" next to the
// executable's rpath / LD_LIBRARY_PATH.  Throws std::runtime_error.
const Api& load(const std::string& path = {});

} // namespace hip
} // namespace acf
