// ref_api_rgbconvert.cpp — extern "C" entry points onto the reference's own `rgbConvert<iT,oT>(I, J, n, d, flag, nrm)`
// (rgbConvertMex.cpp:339-380: the form rgbConvertMex calls at :410) and its rgb2luv / rgb2luv_sse / rgb2gray bodies.
//
// TEST INFRASTRUCTURE ONLY.  No function body of the reference lives here: oracle/Makefile splices lines 9-10 (the
// reference's own wrappers.hpp / sse.hpp includes) and 14-380 of src/lib/acf/acf/toolbox/rgbConvertMex.cpp — everything
// between the OpenCV include at :12 and the cv::Mat-typed wrapper `rgbConvertMex` at :382; that range contains no OpenCV
// token — into a temporary include that exists only while the compiler runs (REF_SPLICE).  Nothing is substituted for
// what is left out.
#include REF_SPLICE

extern "C" {
// flag: 0 gray, 1 rgb, 2 luv, 3 hsv (rgbConvertMex.cpp:389-393); I: d planes of n floats; J: output planes
__attribute__((visibility("default"))) int ref_rgbConvert(float* I, float* J, int n, int d, int flag, float nrm)
{
    try
    {
        rgbConvert<float, float>(I, J, n, d, flag, nrm);
    }
    catch (const char*)
    {
        return 1; // wrError
    }
    return 0;
}
// the scalar body on its own (rgbConvertMex.cpp:62-84), whatever n and the alignment are
__attribute__((visibility("default"))) void ref_rgb2luv_scalar(float* I, float* J, int n, float nrm)
{
    rgb2luv<float, float>(I, J, n, nrm);
}
}
