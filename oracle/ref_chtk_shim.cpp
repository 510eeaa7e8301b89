// oracle/ref_chtk_shim.cpp -- C entry points over the REFERENCE's own HTK reader.
//
// TEST INFRASTRUCTURE ONLY.  This file is the build's; it is compiled TOGETHER WITH the
// reference source /root/reference/chtk/chtk.cpp (taken where it lies, never copied) into
// oracle/_ref/libchtk_ref.so by oracle/Makefile, so that tests can call the reference's
// chtk::load_header (chtk.cpp:97-110) and chtk::htk_load (chtk.cpp:38-88) through ctypes and
// pin plda_amd's HTK path against them bit for bit.
#include <cstring>
#include <string>

#include "chtk.h"

extern "C" {

// out[4] = nsamples, sample_period, samplesize (bytes per frame), parmkind; returns 0, or 1 on error
int ref_htk_header(const char *path, long long out[4]) {
  try {
    chtk::htkheader hd = chtk::load_header(std::string(path));
    out[0] = hd.nsamples; out[1] = hd.sample_period; out[2] = hd.samplesize; out[3] = hd.parmkind;
    return 0;
  } catch (...) {
    return 1;
  }
}

// copies the reference's output (nsamples * (2 frm_ext + 1) * samplesize bytes) into out; returns the
// byte count, 0 if cap is too small (call with cap = 0 to query), -1 on error
long long ref_htk_load(const char *path, int frm_ext, void *out, long long cap) {
  try {
    chtk::htkarray arr = chtk::htk_load(std::string(path), frm_ext);
    const long long bytes = (long long)arr.data_holder->size();
    if (cap < bytes) return cap == 0 ? bytes : 0;
    std::memcpy(out, arr.data_holder->data(), (size_t)bytes);
    return bytes;
  } catch (...) {
    return -1;
  }
}

}  // extern "C"
