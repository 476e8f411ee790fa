// dc_plan.h — partition tree of the tridiagonal divide-and-conquer solver (host side only).
// Every leaf sits at the same depth, so all merges of one height read one eigenvector buffer and
// write the other: the eigenvector matrices ping-pong between two buffers level by level.
#pragma once
#include <vector>

#include "dc_core.h"

namespace psdc {

inline int dc_depth_for(int n) {
  int depth = 0;
  while (((n + (1 << depth) - 1) >> depth) > DC_LEAF) ++depth;
  return depth;
}

// Nodes in depth-first order; *height_out = height of the root (= number of merge levels).
inline void dc_make_plan(int n, std::vector<DcNode>& nodes, int* height_out) {
  const int depth = dc_depth_for(n);
  struct Rec {
    static void split(std::vector<DcNode>& out, int r0, int m, int levels_left) {
      if (levels_left == 0 || m < 2) {
        out.push_back({r0, m, 0, 0});
        // a 1-row node above the leaf level still needs its chain of single-child "merges": avoided
        // by construction (m >= 2^levels_left whenever n >= 2^depth, which dc_depth_for guarantees
        // for n > DC_LEAF; smaller n have depth 0)
        return;
      }
      const int n1 = m / 2;
      out.push_back({r0, m, n1, levels_left});
      split(out, r0, n1, levels_left - 1);
      split(out, r0 + n1, m - n1, levels_left - 1);
    }
  };
  nodes.clear();
  Rec::split(nodes, 0, n, depth);
  *height_out = depth;
}

}  // namespace psdc
