// TEST INFRASTRUCTURE ONLY (tests/emul): the cooperative context switch of the SIMT emulator's fibers, x86-64 System V.
//   void cfd_emul_switch(void** save_sp, void* load_sp): push the callee-saved registers, store rsp to *save_sp, load rsp from
//   load_sp, pop the callee-saved registers, return into the other context.  A new fiber's stack is prepared by
//   cfd_emul::launch (tests/emul/include/hip/hip_runtime.h) to look like one that called this function.
#if !defined(__x86_64__)
#error "the emulator's fiber switch is written for x86-64"
#endif
asm(R"(
    .text
    .globl cfd_emul_switch
    .type cfd_emul_switch, @function
cfd_emul_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size cfd_emul_switch, .-cfd_emul_switch
    .section .note.GNU-stack,"",@progbits
)");
