// ref_api.cpp — extern "C" entry points onto the reference's own toolbox
// functions, so tests can call them through ctypes without C++ name mangling.
//
// TEST INFRASTRUCTURE ONLY.  This file contains declarations and forwarding
// calls only; the function bodies come from the reference sources compiled
// where they lie (see oracle/Makefile: convConst.cpp, gradientMex.cpp,
// wrappers.cpp under /root/reference/src/lib/acf/acf/toolbox/).  It is linked
// into oracle/_ref/libacfref.so, which exists only when /root/reference does.
//
// Prototypes restate the declarations in the reference's OpenCV-typed
// wrappers (convTri.cpp:133-142, gradientMag.cpp:79-82, gradientHist.cpp:85-86).
void convTri(float* I, float* O, int h, int w, int d, int r, int s);
void convTri1(float* I, float* O, int h, int w, int d, float p, int s);
void grad2(float* I, float* Gx, float* Gy, int h, int w, int d);
void gradMag(float* I, float* M, float* O, int h, int w, int d, bool full);
void gradMagNorm(float* M, float* S, int h, int w, float norm);
void gradHist(float* M, float* O, float* H, int h, int w, int bin, int nOrients, int softBin, bool full);

extern "C" {
__attribute__((visibility("default"))) void ref_convTri(float* I, float* O, int h, int w, int d, int r, int s)
{
    convTri(I, O, h, w, d, r, s);
}
__attribute__((visibility("default"))) void ref_convTri1(float* I, float* O, int h, int w, int d, float p, int s)
{
    convTri1(I, O, h, w, d, p, s);
}
__attribute__((visibility("default"))) void ref_grad2(float* I, float* Gx, float* Gy, int h, int w, int d)
{
    grad2(I, Gx, Gy, h, w, d);
}
__attribute__((visibility("default"))) void ref_gradMag(float* I, float* M, float* O, int h, int w, int d, int full)
{
    gradMag(I, M, O, h, w, d, full != 0);
}
__attribute__((visibility("default"))) void ref_gradMagNorm(float* M, float* S, int h, int w, float norm)
{
    gradMagNorm(M, S, h, w, norm);
}
__attribute__((visibility("default"))) void ref_gradHist(float* M, float* O, float* H, int h, int w, int bin, int nOrients, int softBin, int full)
{
    gradHist(M, O, H, h, w, bin, nOrients, softBin, full != 0);
}
}
