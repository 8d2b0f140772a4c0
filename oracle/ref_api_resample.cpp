// ref_api_resample.cpp — extern "C" entry point onto the reference's own `resample<T>` (imResampleMex.cpp:122-383).
//
// TEST INFRASTRUCTURE ONLY.  This file holds NO function body of the reference: oracle/Makefile splices the reference's
// own text, by line range, from the file where it lies under /root/reference into a temporary include that exists only
// while the compiler runs (REF_SPLICE below), and this file forwards to the template it defines.  The ranges taken from
// src/lib/acf/acf/toolbox/imResampleMex.cpp are 9-10 (the reference's own wrappers.hpp / sse.hpp includes), 16-126 and
// 130-383 (std includes, resampleCoef, resample).  Left out: the OpenCV includes at :12-14 and <acf/MatP.h> at :8 (absent
// from this image), the three argument assertions `CV_Assert(A != nullptr / B != nullptr / A != B)` at :127-129 (no
// arithmetic; the callers here never pass null or aliased planes) and the cv::Mat-typed wrapper `imResample` from :385 on.
// Nothing is substituted for what is left out: no stand-in header, macro or type.
#include REF_SPLICE

extern "C" __attribute__((visibility("default"))) void ref_resample(float* A, float* B, int ha, int hb, int wa, int wb, int d, float r)
{
    resample<float>(A, B, ha, hb, wa, wb, d, r);
}
