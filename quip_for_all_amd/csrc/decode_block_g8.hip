// The persistent decode launch of decode_block.hip, compiled a second time for the grouped-query 4096-wide shape (Llama-3-8B,
// Mistral-7B: 32 heads on 8 KV heads, n_ffn = 14336 = 7 x 2048 read as 56 x 256): see QUIP_BLOCK_G8 there.
#define QUIP_BLOCK_G8 1
// (seven gate / up items are in flight at once here: one pre-decoded item each of gate / up and down is what the register file takes)
#ifndef QUIP_PREDECODE_GATE
#define QUIP_PREDECODE_GATE 1
#endif
#ifndef QUIP_PREDECODE_DOWN
#define QUIP_PREDECODE_DOWN 1
#endif
#include "decode_block.hip"
