// Memory for the texts the library hands out (BED, MAF, wig: out_text of include/hgx.h, released with hgx_free): a hundred
// megabytes to gigabytes, written once by many threads.  From malloc such a block is a fresh anonymous mapping of 4 KB pages:
// its first touch is tens of thousands of page faults and its release unmaps them one by one — on a host whose transparent huge
// pages are "madvise" the 119 MB of a lifted million-interval BED cost 7 ms to fill and 15 ms to free, more than lifting it
// (profiles/scripts/r04l_text_timing.py).  Blocks of a megabyte or more are therefore mapped here, advised as huge pages, grown
// with mremap, and one or two released blocks are kept for the next call instead of being unmapped.
#pragma once
#include <cstddef>

namespace hgx {

void *textAlloc(size_t bytes);                      // null when out of memory
void *textRealloc(void *p, size_t bytes);           // p from textAlloc / textRealloc (or null); contents kept; null on failure (p stays)
void textFree(void *p);                             // p from textAlloc / textRealloc, or null
bool textOwns(const void *p);                       // hgx_free: ours, or malloc's?
void textTrim();                                    // unmap what is kept

} // namespace hgx
