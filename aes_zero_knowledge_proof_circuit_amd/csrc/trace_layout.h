// csrc/trace_layout.h -- byte layout of the per-proof AES "trace" buffer shared by the circuit compiler
// (which tags every allocated R1CS variable with the trace bit it equals) and the aes_trace HIP kernel
// (which fills the buffer from message + key).  One trace per chunk-proof:
//
//   [0,16)        key bytes
//   [16,192)      key-schedule words W_0..W_43, 4 bytes each, big-endian byte order (src/aes_circuit.rs:188-212)
//   [192,232)     SubWord(RotWord(W_{i-1})) bytes for i = 4,8,..,40
//   [232,272)     W_{i-4} ^ SubWord(..) (before the Rcon xor) for i = 4,8,..,40
//   [272, ...)    per ECB block, stride TR_BLOCK_STRIDE:
//       +0    message block (16)
//       +16   S_r, r = 0..10 : state after AddRoundKey of round r (S_10 = ciphertext block)     11 x 16
//       +192  SB_r, r = 1..10: state after SubBytes (before ShiftRows)                          10 x 16
//       +352  XT_r, r = 1..9 : xtime ("b") bytes of the ShiftRows output, src/aes_circuit.rs:366-389   9 x 16
//       +496  MP_r, r = 1..9 : for each output byte idx, the 4 partial values of its 5-term xor chain
//                              (src/aes_circuit.rs:391-426); partial 3 is the MixColumns output           9 x 64
#pragma once
#define TR_KEY 0
#define TR_KS_W 16
#define TR_KS_SUB 192
#define TR_KS_PRE 232
#define TR_BLOCK0 272
#define TR_BLOCK_STRIDE 1072
#define TR_BL_MSG 0
#define TR_BL_S 16
#define TR_BL_SB 192
#define TR_BL_XT 352
#define TR_BL_MP 496
#define TR_SBOX_PER_BLOCK 160
#define TR_SBOX_KS 40

// witness descriptors (one u32 per column of z)
#define WD_KIND_SHIFT 30
#define WD_BYTEBIT 0u   // [29:4] trace offset, [3:1] bit, [0] neg
#define WD_SBOX 1u      // [29:11] s-box instance, [10:1] template entry, [0] neg
#define WD_CONST 2u     // [0] value
// s-box template entry: [14:12] level, [11:4] node j at that level, [3:1] bit
