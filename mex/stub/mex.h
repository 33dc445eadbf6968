/* Minimal declaration-only stand-in for MATLAB's mex.h / matrix.h (R2018a interleaved-complex API), just enough to
 * compile mex/isac_mex.cpp in an image without MATLAB (`g++ -fsyntax-only` in __graft_entry__.build(); linked against the test
 * runtime tests/mex_runtime/mx_runtime.cpp for tests/mex_host).  It defines no behaviour; with a real MATLAB the gateway is built
 * against MATLAB's own header:
 *     mex -R2018a mex/isac_mex.cpp -Iinclude -L<package dir> -lisac_hip                                              */
#ifndef ISAC_MEX_STUB_H
#define ISAC_MEX_STUB_H
#include <cstddef>
#include <cstdint>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct { double real, imag; } mxComplexDouble;
typedef double mxDouble;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxLOGICAL_CLASS = 3, mxDOUBLE_CLASS = 6, mxUINT8_CLASS = 9, mxUINT64_CLASS = 15 } mxClassID;

extern "C" {
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
mwSize mxGetNumberOfDimensions(const mxArray*);
const mwSize* mxGetDimensions(const mxArray*);
double mxGetScalar(const mxArray*);
bool mxIsClass(const mxArray*, const char*);
bool mxIsEmpty(const mxArray*);
mxArray* mxGetField(const mxArray*, mwIndex, const char*);
mxArray* mxGetProperty(const mxArray*, mwIndex, const char*);
void mxSetField(mxArray*, mwIndex, const char*, mxArray*);
mxDouble* mxGetDoubles(const mxArray*);
uint64_t* mxGetUint64s(const mxArray*);
mxClassID mxGetClassID(const mxArray*);
bool mxIsStruct(const mxArray*);
mxArray* mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxComplexDouble* mxGetComplexDoubles(const mxArray*);
void* mxGetData(const mxArray*);
char* mxArrayToString(const mxArray*);
void mxFree(void*);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateNumericArray(mwSize, const mwSize*, mxClassID, mxComplexity);
mxArray* mxCreateLogicalMatrix(mwSize, mwSize);
mxArray* mxCreateStructMatrix(mwSize, mwSize, int, const char**);
int mexAtExit(void (*)(void));
void mexErrMsgIdAndTxt(const char*, const char*, ...);
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
}
#endif
