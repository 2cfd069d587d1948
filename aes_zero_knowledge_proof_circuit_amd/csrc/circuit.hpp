// csrc/circuit.hpp -- host-side R1CS compiler for the reference's AES-128-ECB circuit.
//
// Runs once per (circuit kind, block count) inside zkaes_synthesize_keys; it symbolically executes the gates of
// /root/reference/src/lib.rs:60-114,176-293 and src/aes_circuit.rs:20-427 (+ src/helpers/mod.rs:11-64, src/ops.rs:8-29)
// under the ark-r1cs-std 0.3.1 Boolean/UInt8 gadget semantics and emits
//   * A, B, C in CSR form with final column indices (what ark-relations' to_matrices() returns, after the
//     ark-marlin padding: instance padded to a power of two, matrices squared),
//   * one 32-bit *witness descriptor* per column of z saying which bit of the AES trace that variable equals, so that
//     witness generation is a data-parallel gather on the GPU (kernels_witness.hip) rather than a replay of the synthesis.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace zk {

struct CsrMatrix {
    std::vector<uint32_t> rowptr;   // rows + 1
    std::vector<uint32_t> col;
    std::vector<int64_t> coeff;     // small integers (|c| <= 2 for AES; powers of two for ops::add)
    size_t rows() const { return rowptr.empty() ? 0 : rowptr.size() - 1; }
    size_t nnz() const { return col.size(); }
};

enum CircuitKind { CIRCUIT_AES = 0, CIRCUIT_OPS_XOR = 1, CIRCUIT_OPS_ADD = 2 };

struct Circuit {
    int kind = CIRCUIT_AES;
    size_t n_blocks = 0;
    // before padding (what debug_constraint_system_status would log, src/helpers/mod.rs:73-81)
    size_t raw_constraints = 0, raw_instance = 0, raw_witness = 0;
    // after ark-marlin padding
    size_t num_instance = 0, num_witness = 0, num_constraints = 0;
    CsrMatrix A, B, C;
    std::vector<uint32_t> desc;          // num_instance + num_witness descriptors (trace_layout.h)
    std::vector<uint32_t> sbox_in_off;   // trace offset of the input byte of every S-box instance
    std::vector<uint32_t> sbox_tmpl;     // (level, node, bit) of every allocated variable of one S-box
    size_t trace_bytes = 0;
    size_t num_variables() const { return num_instance + num_witness; }
};

// message_len must be a multiple of 16 (else throws std::invalid_argument with the reference's message)
Circuit compile_aes_circuit(size_t message_len);
Circuit compile_ops_circuit(int kind);
uint8_t aes_sbox_value(uint8_t x);   // the lookup table of src/aes_circuit.rs:433-694

}  // namespace zk
