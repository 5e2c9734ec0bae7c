// Host decisions of the mask refinement (csrc/host_refine.cpp), shared with csrc/tail.hip.
#pragma once
#include <stdint.h>

struct RRule {       // candidate rule of a window
  int32_t kind;      // -1 unused; 0 grey range [lo, hi] (integer bounds of cv2.inRange); 1..3 channel B/G/R > lo
  int32_t lo, hi;
};
struct RCand {       // a chosen candidate: rule index 0..5, polarity, xor distance
  int rule;
  int invert;
  unsigned long long dist;
};
void refine_rules(const uint32_t* hist4 /* (4,256) */, RRule rules[6]);
int refine_candidates(const RRule rules[6], const uint64_t sums[6], long long npix, RCand out[4]);
