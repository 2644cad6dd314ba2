// tests/emu/emu_lds.cpp -- storage behind the kernels' dynamic-LDS declarations (`extern __shared__ T name[]`), one
// definition per name used in miniasm_amd/csrc.  TEST INFRASTRUCTURE ONLY.
// paf.hip: k_paf_parse stages the text of its lines in s_text
alignas(16) thread_local unsigned char s_text[160 << 10];
