/* tests/stubs/mex.h -- declaration-only stand-in for MATLAB's mex.h, used ONLY by
 * tests/test_host.py::test_mex_gateway_compiles to syntax- and type-check
 * em_model_manned_bayes_amd/matlab/emgpu_mex.c with gcc -fsyntax-only.  Nothing here is ever linked or run;
 * a real build uses MATLAB's own header. */
#ifndef EMGPU_TEST_MEX_STUB_H
#define EMGPU_TEST_MEX_STUB_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxDOUBLE_CLASS = 6, mxUINT64_CLASS = 13, mxUINT8_CLASS = 9 } mxClassID;
typedef enum { mxREAL = 0 } mxComplexity;
int mxGetString(const mxArray *a, char *buf, mwSize n);
void *mxGetData(const mxArray *a);
double *mxGetPr(const mxArray *a);
double mxGetScalar(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
int mxIsChar(const mxArray *a);
int mxIsEmpty(const mxArray *a);
int mxIsLogicalScalarTrue(const mxArray *a);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity f);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity f);
mxArray *mxCreateNumericArray(mwSize nd, const mwSize *dims, mxClassID c, mxComplexity f);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateLogicalMatrix(mwSize m, mwSize n);
mxArray *mxCreateCellMatrix(mwSize m, mwSize n);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames);
mxArray *mxCreateString(const char *s);
mxArray *mxGetCell(const mxArray *a, mwSize i);
void mxSetCell(mxArray *a, mwSize i, mxArray *v);
mxArray *mxGetField(const mxArray *a, mwSize i, const char *name);
void mxSetField(mxArray *a, mwSize i, const char *name, mxArray *v);
unsigned char *mxGetLogicals(const mxArray *a);
int mxIsLogical(const mxArray *a);
int mxIsDouble(const mxArray *a);
int mxIsCell(const mxArray *a);
int mxIsStruct(const mxArray *a);
int mexAtExit(void (*fn)(void));
void *mxMalloc(size_t n);
void *mxCalloc(size_t n, size_t size);
void mxFree(void *p);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
#endif
