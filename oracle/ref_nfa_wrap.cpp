// oracle/_ref: Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp - the only text of LSD's nfa() / log_gamma() arithmetic inside the
// reference tree (the LSD the reference LINKS is OpenCV's lsd.cpp, which is not in the tree; this file is its in-tree twin, used
// by the vendored EDLines code).  Built from the sources where they lie under /root/reference (oracle/Makefile target `_ref`;
// nothing of the reference is copied into this repository).  This wrapper is ours: C entry points around the class's private
// members so that the CPU suite can compare oracle/line_oracle.cpp's nfa() / log_gamma() with the reference's own code.
// Test infrastructure only.
#define private public   // NFALUT::nfa and NFALUT::log_gamma are private members; the class layout does not depend on access
#include "NFA.h"
#undef private

extern "C" double ref_nfa(int n, int k, double p, double logNT) {
    NFALUT lut(1, p, logNT);   // a look-up table of one entry: the constructor evaluates nothing
    return lut.nfa(n, k);
}

extern "C" double ref_log_gamma(double x) { return NFALUT::log_gamma(x); }
