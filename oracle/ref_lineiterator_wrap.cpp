// oracle/_ref: the ONE file of the reference's hot path that compiles without OpenCV / Eigen, add_src/lineIterator.cpp
// (Bresenham walk behind Frame::AssignFeaturesToGridForLine, src/Frame.cc:286-309), built from the sources where they lie
// under /root/reference (oracle/Makefile target `_ref`; nothing of the reference is copied into this repository).  This
// wrapper is ours: a C entry point around ORB_SLAM2::LineIterator so that the CPU suite can compare the oracle's restated walk
// (oracle/linematch_oracle.cpp: LineIt) with the reference's own code.  Test infrastructure only.
#include <utility>

#include "lineIterator.h"

extern "C" int ref_line_iterator_walk(double x1, double y1, double x2, double y2, int* xy, int cap) {
    ORB_SLAM2::LineIterator it(x1, y1, x2, y2);
    std::pair<int, int> p;
    int n = 0;
    while (it.getNext(p)) {
        if (n < cap) { xy[2 * n] = p.first; xy[2 * n + 1] = p.second; }
        ++n;
    }
    return n;
}
