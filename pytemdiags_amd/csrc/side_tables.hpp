// side_tables.hpp -- host: the row table of the latitude-class sweeps (kernels_cls.hpp: crow, one batch of 4 member
// rows for each of the 4 class slots of a class-group, northern batches first) split into one table per side.
// sweep_os2_kernel (kernels_op2.hpp) gives every class side a wave of its own; a wave then walks ITS table.
// Same entry format (row | flags << 28 | has-padding << 27, padding entries negative); CLS_FIRST / CLS_LAST mark the
// first / last batch of the group ON THAT SIDE; a group without members on a side gets one batch of padding
// entries, so both tables have every group.  gfirst[g] = first batch of group g, gfirst[ngroups] = batches.
#pragma once
#include <cstdint>
#include <vector>

namespace temx {

struct SideTables {
  std::vector<int> crow[2];     // [batches + CLS_PADB][4][CLS_MB]
  std::vector<int> gfirst[2];   // [ngroups + 1]
};

inline void build_side_tables(const std::vector<int>& crow, const std::vector<int>& gbatch0, int64_t ngroups, int mb,
                              int padb, int south_flag, int first_flag, int last_flag, int haspad_bit, SideTables& out) {
  const int per = 4 * mb;
  for (int side = 0; side < 2; ++side) {
    out.crow[side].clear();
    out.gfirst[side].assign((size_t)ngroups + 1, 0);
  }
  for (int64_t g = 0; g < ngroups; ++g) {
    std::vector<int> mine[2];
    for (int b = gbatch0[(size_t)g]; b < gbatch0[(size_t)g + 1]; ++b) {
      const int fl = (crow[(size_t)b * per] >> 28) & 7;
      mine[(fl & south_flag) ? 1 : 0].push_back(b);
    }
    for (int side = 0; side < 2; ++side) {
      std::vector<int>& t = out.crow[side];
      out.gfirst[side][(size_t)g] = (int)(t.size() / per);
      const size_t nb = mine[side].size();
      if (nb == 0) {
        const int fl = first_flag | last_flag;
        for (int e = 0; e < per; ++e) t.push_back((int)0x80000000 | (fl << 28) | haspad_bit);
        continue;
      }
      for (size_t i = 0; i < nb; ++i) {
        const int fl = (i == 0 ? first_flag : 0) | (i + 1 == nb ? last_flag : 0);
        for (int e = 0; e < per; ++e) {
          const int ent = crow[(size_t)mine[side][i] * per + e];
          t.push_back((ent & ~(7 << 28)) | (fl << 28));        // row, padding sign and has-padding bit stay
        }
      }
    }
  }
  for (int side = 0; side < 2; ++side) {
    out.gfirst[side][(size_t)ngroups] = (int)(out.crow[side].size() / per);
    out.crow[side].resize(out.crow[side].size() + (size_t)padb * per, (int)0x80000000);
  }
}

}  // namespace temx
