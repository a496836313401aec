// Host test of go-snark-study_amd/csrc/knobs.h: compiled twice by tests/test_host_logic.py, with and without -DGS_DEV_KNOBS.
// Reads lines "<kind> <NAME> <default> <lo> <hi> <multiple_of> <pow2>" and prints the value the library would use.
#include <cstdio>
#include <cstring>

#include "knobs.h"

int main() {
  char kind[32], name[64];
  long d, lo, hi, mult;
  int pow2;
  while (scanf("%31s %63s %ld %ld %ld %ld %d", kind, name, &d, &lo, &hi, &mult, &pow2) == 7) {
    if (!strcmp(kind, "dev")) printf("%ld\n", gs::dev_knob(name, d, lo, hi, mult, pow2 != 0));
    else if (!strcmp(kind, "run")) printf("%ld\n", gs::run_knob(name, d, lo, hi));
    else if (!strcmp(kind, "devflag")) printf("%d\n", gs::dev_flag(name) ? 1 : 0);
    else printf("?\n");
  }
  return 0;
}
