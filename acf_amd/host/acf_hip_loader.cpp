#include "acf_hip_loader.h"

#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <stdexcept>

namespace acf
{
namespace hip
{

static Api g_api;
static std::once_flag g_once;
static std::string g_error;

static void doLoad(const std::string& path)
{
    std::string p = path;
    if (p.empty())
    {
        const char* env = std::getenv("ACF_HIP_LIBRARY");
        p = env ? env : "libacf_hip.so";
    }
    void* h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h)
    {
        g_error = std::string("acf::hip::load: cannot dlopen ") + p + ": " + dlerror();
        return;
    }
    g_api.handle = h;
#define ACF_HIP_FN(name)                                                                  \
    g_api.name = reinterpret_cast<decltype(g_api.name)>(dlsym(h, #name));                 \
    if (!g_api.name)                                                                      \
    {                                                                                     \
        g_error = std::string("acf::hip::load: ") + p + " does not export " #name;        \
        return;                                                                           \
    }
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
    if (g_api.acf_hip_abi_version() != ACF_HIP_ABI_VERSION)
    {
        g_error = "acf::hip::load: ABI version mismatch between acf_hip.h and " + p;
    }
}

const Api& load(const std::string& path)
{
    std::call_once(g_once, doLoad, path);
    if (!g_error.empty())
    {
        throw std::runtime_error(g_error);
    }
    return g_api;
}

} // namespace hip
} // namespace acf
