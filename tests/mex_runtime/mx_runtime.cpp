// A small in-process implementation of the MATLAB MEX / mx API subset that mex/isac_mex.cpp uses (declared in mex/stub/mex.h), so that the
// gateway can be LINKED AND RUN here without MATLAB: tests/mex_host.cpp builds the argument lists a MATLAB caller would pass
// (structs, value objects with properties, char vectors, interleaved-complex arrays, uint64 handles), calls mexFunction() and reads
// the results back.  Test infrastructure only -- with a real MATLAB the gateway is built by `mex -R2018a` against MATLAB's own runtime.
// Semantics follow the documented behaviour of the functions (column-major data, mxGetN = product of the trailing dimensions,
// mxGetField / mxGetProperty return NULL for unknown names, mexErrMsgIdAndTxt does not return: it throws MexError here).
#include "mx_runtime.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct mxArray_tag {
  mxClassID cls = mxDOUBLE_CLASS;
  bool cplx = false, is_struct = false, is_object = false, is_char = false;
  std::string class_name = "double";
  std::vector<mwSize> dims{0, 0};
  std::vector<unsigned char> data;
  std::vector<std::string> field_order;
  std::map<std::string, mxArray*> fields;        // struct fields / object properties of element 0 (only 1x1 structs are used)
  std::string str;
};

namespace {
std::vector<void (*)(void)> g_at_exit;
size_t elem_bytes(mxClassID c, bool cplx) {
  size_t b = 8;
  if (c == mxLOGICAL_CLASS || c == mxUINT8_CLASS) b = 1;
  if ((int)c == 12 || (int)c == 13) b = 4;          // mxINT32_CLASS / mxUINT32_CLASS
  return cplx ? 2 * b : b;
}
size_t numel(const mxArray* a) {
  size_t n = 1;
  for (mwSize d : a->dims) n *= d;
  return n;
}
mxArray* make(mxClassID c, bool cplx, const std::vector<mwSize>& dims, const char* cname) {
  mxArray* a = new mxArray_tag;
  a->cls = c; a->cplx = cplx; a->dims = dims; a->class_name = cname;
  a->data.assign(numel(a) * elem_bytes(c, cplx), 0);
  return a;
}
}  // namespace

MexError::MexError(std::string id_, std::string msg_) : std::runtime_error(id_ + ": " + msg_), id(std::move(id_)), msg(std::move(msg_)) {}

// ---- helpers for the test harness (not part of the MEX API)
mxArray* mxr_string(const char* s) {
  mxArray* a = new mxArray_tag;
  a->is_char = true; a->class_name = "char"; a->str = s; a->dims = {1, (mwSize)std::strlen(s)};
  return a;
}
mxArray* mxr_object(const char* class_name) {
  mxArray* a = new mxArray_tag;
  a->is_object = true; a->class_name = class_name; a->dims = {1, 1};
  return a;
}
void mxr_set_property(mxArray* obj, const char* name, mxArray* v) { obj->fields[name] = v; }
mxArray* mxr_empty() { return make(mxDOUBLE_CLASS, false, {0, 0}, "double"); }
mxArray* mxr_uint8(const unsigned char* v, mwSize n) {
  mxArray* a = make(mxUINT8_CLASS, false, {1, n}, "uint8");
  std::memcpy(a->data.data(), v, n);
  return a;
}
mxArray* mxr_int32(const int32_t* v, mwSize n) {
  mxArray* a = make((mxClassID)12, false, {1, n}, "int32");
  std::memcpy(a->data.data(), v, sizeof(int32_t) * n);
  return a;
}
void mxr_run_at_exit() {
  for (auto f : g_at_exit) f();
  g_at_exit.clear();
}
void mxr_destroy(mxArray* a) {
  if (!a) return;
  for (auto& kv : a->fields) mxr_destroy(kv.second);
  delete a;
}

// ---- the API subset
extern "C" {
size_t mxGetM(const mxArray* a) { return a->dims.empty() ? 0 : a->dims[0]; }
size_t mxGetN(const mxArray* a) {
  size_t n = 1;
  for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i];
  return a->dims.size() < 2 ? 1 : n;
}
size_t mxGetNumberOfElements(const mxArray* a) { return numel(a); }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return (mwSize)a->dims.size(); }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims.data(); }
double mxGetScalar(const mxArray* a) {
  if (numel(a) == 0) mexErrMsgIdAndTxt("MATLAB:mxGetScalar", "empty array");
  switch (a->cls) {
    case mxDOUBLE_CLASS: return *reinterpret_cast<const double*>(a->data.data());
    case mxUINT64_CLASS: return (double)*reinterpret_cast<const uint64_t*>(a->data.data());
    default: return (double)a->data[0];
  }
}
bool mxIsClass(const mxArray* a, const char* name) { return a->class_name == name; }
bool mxIsEmpty(const mxArray* a) { return numel(a) == 0; }
bool mxIsStruct(const mxArray* a) { return a->is_struct; }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
mxArray* mxGetField(const mxArray* a, mwIndex i, const char* name) {
  if (!a->is_struct || i != 0) return nullptr;
  auto it = a->fields.find(name);
  return it == a->fields.end() ? nullptr : it->second;
}
mxArray* mxGetProperty(const mxArray* a, mwIndex i, const char* name) {
  if (!a->is_object || i != 0) return nullptr;
  auto it = a->fields.find(name);
  return it == a->fields.end() ? nullptr : it->second;
}
void mxSetField(mxArray* a, mwIndex i, const char* name, mxArray* v) {
  if (!a->is_struct || i != 0) mexErrMsgIdAndTxt("MATLAB:mxSetField", "not a 1x1 struct");
  auto it = a->fields.find(name);
  if (it == a->fields.end()) mexErrMsgIdAndTxt("MATLAB:mxSetField", "no such field %s", name);
  mxr_destroy(it->second);
  it->second = v;
}
mxDouble* mxGetDoubles(const mxArray* a) {
  if (a->cls != mxDOUBLE_CLASS || a->cplx || a->is_struct || a->is_object || a->is_char) mexErrMsgIdAndTxt("MATLAB:mxGetDoubles", "not a real double array");
  return reinterpret_cast<mxDouble*>(const_cast<unsigned char*>(a->data.data()));
}
mxComplexDouble* mxGetComplexDoubles(const mxArray* a) {
  if (a->cls != mxDOUBLE_CLASS || !a->cplx) mexErrMsgIdAndTxt("MATLAB:mxGetComplexDoubles", "not a complex double array");
  return reinterpret_cast<mxComplexDouble*>(const_cast<unsigned char*>(a->data.data()));
}
uint64_t* mxGetUint64s(const mxArray* a) {
  if (a->cls != mxUINT64_CLASS) mexErrMsgIdAndTxt("MATLAB:mxGetUint64s", "not a uint64 array");
  return reinterpret_cast<uint64_t*>(const_cast<unsigned char*>(a->data.data()));
}
void* mxGetData(const mxArray* a) { return const_cast<unsigned char*>(a->data.data()); }
char* mxArrayToString(const mxArray* a) {
  if (!a->is_char) return nullptr;
  char* s = (char*)std::malloc(a->str.size() + 1);
  std::memcpy(s, a->str.c_str(), a->str.size() + 1);
  return s;
}
void mxFree(void* p) { std::free(p); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity k) {
  return make(c, k == mxCOMPLEX, {m, n}, c == mxUINT64_CLASS ? "uint64" : c == mxUINT8_CLASS ? "uint8" : "double");
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity k) { return make(mxDOUBLE_CLASS, k == mxCOMPLEX, {m, n}, "double"); }
mxArray* mxCreateDoubleScalar(double v) {
  mxArray* a = make(mxDOUBLE_CLASS, false, {1, 1}, "double");
  *reinterpret_cast<double*>(a->data.data()) = v;
  return a;
}
mxArray* mxCreateNumericArray(mwSize nd, const mwSize* d, mxClassID c, mxComplexity k) {
  std::vector<mwSize> dims(d, d + nd);
  while (dims.size() > 2 && dims.back() == 1) dims.pop_back();      // MATLAB drops trailing singleton dimensions
  return make(c, k == mxCOMPLEX, dims, "double");
}
mxArray* mxCreateLogicalMatrix(mwSize m, mwSize n) { return make(mxLOGICAL_CLASS, false, {m, n}, "logical"); }
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nf, const char** names) {
  mxArray* a = new mxArray_tag;
  a->is_struct = true; a->class_name = "struct"; a->dims = {m, n};
  for (int i = 0; i < nf; ++i) { a->field_order.push_back(names[i]); a->fields[names[i]] = mxr_empty(); }
  return a;
}
int mexAtExit(void (*f)(void)) { g_at_exit.push_back(f); return 0; }
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw MexError(id, buf);
}
}
