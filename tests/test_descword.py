"""The two words of an interior descriptor (svo-raytracer_amd/csrc/svo_descword.h), compiled for the host: every combination of
empty / leaf / descendable children.  What the trips rely on (svo_travloop2.h, svo_trav2.h::trav_step2): a child's nibble is 0
iff the child is empty, >= 8 iff the walk can descend into it, and desc.x + 8 * nibble is then the byte offset of the child's
descriptor -- the rank-th of the group that starts at the first child's index."""
import itertools
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
#include <cstdio>
#include "svo_descword.h"
using namespace svo::derive;
int main() {
  for (uint32_t ne = 0; ne < 256; ne++)
    for (uint32_t has = 0; has < 256; has++) {
      if (has & ~ne) continue;
      std::printf("%u %u %u %u\n", ne, has, desc_word(ne, has), desc_has(desc_word(ne, has)));
    }
  const uint32_t firsts[] = {0u, 1u, 2u, 7u, 8u, 9u, 12345u, (1u << 28) - 9u};
  for (uint32_t f : firsts) std::printf("F %u %u %u\n", f, desc_base(f), desc_first(desc_base(f)));
  return 0;
}
"""


def test_every_combination_of_children(tmp_path):
    src = tmp_path / "descword.cpp"
    src.write_text(PROG)
    exe = tmp_path / "descword"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "svo-raytracer_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    rows = [tuple(int(x) for x in ln.split()) for ln in out if not ln.startswith("F")]
    assert len(rows) == 3 ** 8          # per child: empty, not empty, not empty with a child block
    for ne, has, word, has_back in rows:
        assert has_back == has
        rank = 0
        for c in range(8):
            nib = (word >> (4 * c)) & 15
            if (has >> c) & 1:
                assert nib == (8 | rank)
                rank += 1
            elif (ne >> c) & 1:
                assert nib == 1
            else:
                assert nib == 0
    for ln in out:
        if ln.startswith("F"):
            _, first, base, back = ln.split()
            first, base, back = int(first), int(base), int(back)
            assert back == first
            for rank in range(8):       # what a DESCEND computes: desc.x + 8 * (8 | rank), in 32 bits
                assert (base + 8 * (8 | rank)) & 0xffffffff == 8 * (first + rank)
