// TEST INFRASTRUCTURE ONLY.  C-ABI shim around the reference's own CTW implementation so that tests can call the
// real reference (compiled from /root/reference/chaos/cppctw.cpp where it lies - never copied into this repo; see
// oracle/Makefile).  Mirrors the call chain ctw.pyx:2 -> cppctw.hpp:21 -> cppctw.cpp:160-171.
#include <vector>

#include "cppctw.h"  // -I/root/reference/chaos

extern "C" double ref_ctw_estimate_entropy(const char* seq, long n, int alphabet_size) {
  std::vector<char> v(seq, seq + n);
  return estimate_entropy(v, (char)alphabet_size);
}
