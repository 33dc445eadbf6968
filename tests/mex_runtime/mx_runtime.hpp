// Test-side additions to mex/stub/mex.h: the exception mexErrMsgIdAndTxt raises in the in-process runtime (tests/mex_runtime/mx_runtime.cpp)
// and constructors for the argument kinds the MEX C API itself cannot create (char vectors, value objects with properties).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

#include "mex.h"

struct MexError : std::runtime_error {
  std::string id, msg;
  MexError(std::string id_, std::string msg_);
};

mxArray* mxr_string(const char* s);                                  // 'text'
mxArray* mxr_object(const char* class_name);                         // a value object: mxIsClass(obj, class_name), properties via mxGetProperty
void mxr_set_property(mxArray* obj, const char* name, mxArray* v);   // (takes ownership of v)
mxArray* mxr_empty();                                                // []
mxArray* mxr_uint8(const unsigned char* v, mwSize n);                // uint8 row vector
mxArray* mxr_int32(const int32_t* v, mwSize n);                      // int32 row vector
void mxr_run_at_exit();                                              // what MATLAB does when the MEX file is cleared
void mxr_destroy(mxArray* a);
