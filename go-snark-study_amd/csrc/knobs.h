// Environment switches of the library, in ONE place (round 3 shipped ~15 raw getenv/atoi calls, one of
// which -- GS_REDUCE_L=3 -- silently produced a wrong MSM).
//
//  * run_flag / run_knob: available in every build.  They select between code paths that give the SAME results (stream layout,
//    tracing, copy threads); numeric ones are parsed strictly and a value outside their range is ignored (the default stands).
//  * dev_knob / dev_knob_f / dev_flag: tuning of the algorithms themselves (window model, chunk sizes, reduce shape, ...).
//    Compiled OUT of the product library: they read the environment only in a development build (`make EXTRA=-DGS_DEV_KNOBS ...`,
//    how tools/gpu_run.sh's A/B variants are made), and even there a value the algorithm cannot take -- out of range, not a
//    multiple of `multiple_of`, not a power of two where one is needed -- falls back to the default instead of being used.
#pragma once
#include <cerrno>
#include <cstdlib>

namespace gs {

inline bool knob_parse(const char* name, long lo, long hi, long multiple_of, bool pow2, long& out) {
  const char* s = getenv(name);
  if (!s || !*s) return false;
  char* end = nullptr;
  errno = 0;
  const long v = strtol(s, &end, 10);
  if (errno != 0 || end == s || *end != '\0') return false;
  if (v < lo || v > hi) return false;
  if (multiple_of > 1 && v % multiple_of != 0) return false;
  if (pow2 && (v <= 0 || (v & (v - 1)) != 0)) return false;
  out = v;
  return true;
}

inline bool run_flag(const char* name) { return getenv(name) != nullptr; }
inline long run_knob(const char* name, long dflt, long lo, long hi) {
  long v = dflt;
  return knob_parse(name, lo, hi, 1, false, v) ? v : dflt;
}

#ifdef GS_DEV_KNOBS
inline bool dev_flag(const char* name) { return getenv(name) != nullptr; }
inline long dev_knob(const char* name, long dflt, long lo, long hi, long multiple_of = 1, bool pow2 = false) {
  long v = dflt;
  return knob_parse(name, lo, hi, multiple_of, pow2, v) ? v : dflt;
}
inline double dev_knob_f(const char* name, double dflt, double lo, double hi) {
  const char* s = getenv(name);
  if (!s || !*s) return dflt;
  char* end = nullptr;
  const double v = strtod(s, &end);
  return (end == s || *end != '\0' || !(v >= lo && v <= hi)) ? dflt : v;
}
#else
inline bool dev_flag(const char*) { return false; }
inline long dev_knob(const char*, long dflt, long, long, long = 1, bool = false) { return dflt; }
inline double dev_knob_f(const char*, double dflt, double, double) { return dflt; }
#endif

}  // namespace gs
