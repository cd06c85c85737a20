/* test shim: exposes the header-only corpus generator (zarc_amd/csrc/corpus.h) to ctypes */
#include "../../zarc_amd/csrc/corpus.h"
void corpus_entry(uint8_t *dst, size_t n, uint64_t index, int kind) { zarc_corpus_entry(dst, n, index, kind); }
