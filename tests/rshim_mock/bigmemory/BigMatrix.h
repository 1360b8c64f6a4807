// Stand-in for the part of bigmemory's BigMatrix interface the shim uses (nrow / ncol / matrix): TEST
// INFRASTRUCTURE ONLY, see tests/rshim_mock/Rcpp.h.
#pragma once
typedef long index_type;
class BigMatrix {
 public:
  index_type nrow() const;
  index_type ncol() const;
  void* matrix();
};
