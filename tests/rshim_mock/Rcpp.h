// Minimal stand-in for the declarations of <Rcpp.h> / R's C API that r-shim/src/bigkrls_shim.cpp uses -- TEST
// INFRASTRUCTURE ONLY (tests/test_rshim_cpu.py type-checks the shim against it with g++ -fsyntax-only; R and Rcpp
// are absent from the build image). Declarations follow the documented public interfaces (Rcpp's vectors / List /
// XPtr / Nullable / stop, "Writing R Extensions" section 5.13 for external pointers); nothing here is ever linked
// or run, and nothing of the product is built with it.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <initializer_list>
#include <string>
#include <vector>

struct SEXPREC;
typedef SEXPREC* SEXP;
extern SEXP R_NilValue;
typedef int Rboolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define EXTPTRSXP 22
typedef void (*R_CFinalizer_t)(SEXP);
extern "C" {
int TYPEOF(SEXP);
int Rf_isNull(SEXP);
void* R_ExternalPtrAddr(SEXP);
SEXP R_ExternalPtrProtected(SEXP);
SEXP R_MakeExternalPtr(void*, SEXP tag, SEXP prot);
void R_ClearExternalPtr(SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean onexit);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
}
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)

namespace Rcpp {

template <class T>
class XPtr {                                   // XPtr<T>(SEXP): typed view of an external pointer
 public:
  explicit XPtr(SEXP s) : p_((T*)R_ExternalPtrAddr(s)), s_(s) {}
  T* operator->() const { return p_; }
  T* get() const { return p_; }
  operator T*() const { return p_; }
  operator SEXP() const { return s_; }
 private:
  T* p_;
  SEXP s_;
};

template <class E>
class VectorMock {
 public:
  VectorMock() {}
  VectorMock(long n) : v_((size_t)n) {}
  VectorMock(int n) : v_((size_t)n) {}
  VectorMock(SEXP) {}
  typedef E* iterator;
  E* begin() { return v_.data(); }
  const E* begin() const { return v_.data(); }
  E* end() { return v_.data() + v_.size(); }
  const E* end() const { return v_.data() + v_.size(); }
  long size() const { return (long)v_.size(); }
  E& operator[](long i) { return v_[(size_t)i]; }
  operator SEXP() const { return nullptr; }
  static VectorMock create(E a, E b) { VectorMock r(2); r[0] = a; r[1] = b; return r; }
 private:
  std::vector<E> v_;
};
typedef VectorMock<double> NumericVector;
typedef VectorMock<int> IntegerVector;
typedef VectorMock<unsigned char> RawVector;

class NumericMatrix {
 public:
  NumericMatrix(long nrow, long ncol) : r_(nrow), c_(ncol), v_((size_t)(nrow * ncol)) {}
  NumericMatrix(SEXP) : r_(0), c_(0) {}
  double* begin() { return v_.data(); }
  const double* begin() const { return v_.data(); }
  long nrow() const { return r_; }
  long ncol() const { return c_; }
  operator SEXP() const { return nullptr; }
 private:
  long r_, c_;
  std::vector<double> v_;
};

template <class T>
class Nullable {
 public:
  Nullable() : s_(nullptr) {}
  Nullable(SEXP s) : s_(s) {}
  bool isNotNull() const { return s_ != nullptr; }
  bool isNull() const { return s_ == nullptr; }
  operator SEXP() const { return s_; }
 private:
  SEXP s_;
};

struct NamedArg { const char* name; };
struct NameProxy {
  const char* name;
  template <class V> NamedArg operator=(const V&) const { return NamedArg{name}; }
};
struct Placeholder { NameProxy operator[](const char* nm) const { return NameProxy{nm}; } };
static Placeholder _;

class List {
 public:
  template <class... A> static List create(const A&...) { return List(); }
  operator SEXP() const { return nullptr; }
};

[[noreturn]] void stop(const char* fmt, ...);

}  // namespace Rcpp
