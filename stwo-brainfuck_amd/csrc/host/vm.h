// Host side of the drop-in: Brainfuck compiler + VM producing the execution trace the prover consumes
// (`Machine` argument of prove_brainfuck, crates/brainfuck_prover/src/brainfuck_air/mod.rs:471). Sequential interpreter, so it
// stays on the CPU (out of GPU scope). Semantics: crates/brainfuck_vm/src/compiler.rs:17-37,
// crates/brainfuck_vm/src/machine.rs:141-238.
#pragma once
#include "../air.h"
#include <vector>
#include <string>
#include <stdexcept>

namespace bf {

// crates/brainfuck_vm/src/registers.rs:6-21
struct Registers { u32 clk = 0, ip = 0, ci = 0, ni = 0, mp = 0, mv = 0, mvi = 0; };

// crates/brainfuck_vm/src/instruction.rs:65-76 — opcodes are the ASCII codes.
// opcodes: see air.h (OP_*), ASCII codes as in crates/brainfuck_vm/src/instruction.rs:65-76

// compiler.rs:13-37: strip whitespace; each symbol -> its code; '[' and ']' are followed by a jump-target word.
static inline std::vector<u32> compile(const std::string& code) {
    std::vector<u32> ins;
    std::vector<size_t> loop_stack;
    for (unsigned char c : code) {
        if (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f') continue;
        ins.push_back((u32)c);
        if (c == '[') { ins.push_back(0); loop_stack.push_back(ins.size() - 1); }
        else if (c == ']') {
            if (loop_stack.empty()) throw std::runtime_error("unbalanced ]");
            size_t start = loop_stack.back(); loop_stack.pop_back();
            ins[start] = (u32)ins.size();
            ins.push_back((u32)(start + 1));
        }
    }
    return ins;
}

struct Machine {
    std::vector<u32> code;
    std::vector<u32> ram;
    std::vector<u8> input; size_t in_pos = 0;
    std::vector<u8> output;
    Registers reg;
    std::vector<Registers> trace;
    std::vector<u32> inv_memo = std::vector<u32>(1 << 16, 0);   // mv -> mv^-1 for small cell values (same value as m_inv, computed once)
    static constexpr size_t DEFAULT_RAM_SIZE = 30000;  // machine.rs:114

    Machine(std::vector<u32> code_, std::vector<u8> input_, size_t ram_size = DEFAULT_RAM_SIZE)
        : code(std::move(code_)), ram(ram_size, 0), input(std::move(input_)) {}

    // machine.rs:141-161
    void execute() {
        trace.reserve(1 << 16);
        while (reg.ip < code.size()) {
            reg.ci = code[reg.ip];
            reg.ni = (reg.ip == code.size() - 1) ? 0 : code[reg.ip + 1];
            trace.push_back(reg);
            step(reg.ci);
            reg.clk = m_add(reg.clk, 1);  // machine.rs:231-234
            reg.ip = m_add(reg.ip, 1);
        }
        reg.ci = 0; reg.ni = 0;
        trace.push_back(reg);
    }

   private:
    // machine.rs:177-229
    void step(u32 ins) {
        switch (ins) {
            case OP_RIGHT: reg.mp = m_add(reg.mp, 1); break;
            case OP_LEFT: reg.mp = m_sub(reg.mp, 1); break;
            case OP_PLUS: ram.at(reg.mp) = m_add(ram.at(reg.mp), 1); break;
            case OP_MINUS: ram.at(reg.mp) = m_sub(ram.at(reg.mp), 1); break;
            case OP_READCHAR:
                if (in_pos >= input.size()) throw std::runtime_error("input exhausted");
                ram.at(reg.mp) = input[in_pos++];
                break;
            case OP_PUTCHAR: output.push_back((u8)ram.at(reg.mp)); break;
            case OP_JZ: {
                u32 arg = code.at(reg.ip + 1);
                reg.ni = arg;
                if (ram.at(reg.mp) == 0) { reg.ip = arg; return; }
                reg.ip = m_add(reg.ip, 1);
                break;
            }
            case OP_JNZ: {
                u32 arg = code.at(reg.ip + 1);
                if (ram.at(reg.mp) != 0) { reg.ip = m_sub(arg, 1); return; }
                reg.ip = m_add(reg.ip, 1);
                break;
            }
            default: throw std::runtime_error("invalid instruction");
        }
        reg.mv = ram.at(reg.mp);
        if (reg.mv == 0) reg.mvi = 0;
        else if (reg.mv < inv_memo.size()) { u32& slot = inv_memo[reg.mv]; if (!slot) slot = m_inv(reg.mv); reg.mvi = slot; }
        else reg.mvi = m_inv(reg.mv);
    }
};

}  // namespace bf
