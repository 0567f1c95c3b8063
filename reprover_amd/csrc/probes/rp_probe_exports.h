// PROBE BUILDS ONLY (-DRP_PHASE_PROBE): copy the per-workgroup phase timestamps (probes/rp_probe_hooks.h) to the host.
// Every translation unit that includes rp_gemm.h holds its own copy of the timestamp arrays; RP_PROBE_EXPORT_SCAN selects
// the readers of rp_retrieval.hip's copy (the similarity scan's filter pass), the default those of rp_encoder.hip's.
#pragma once
#ifdef RP_PROBE_EXPORT_SCAN
extern "C" int rp_probe_read_scan_phase_ts(unsigned long long* host_out, int n_words) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rp::g_phase_ts), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#else
extern "C" int rp_probe_read_phase_ts(unsigned long long* host_out, int n_words) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rp::g_phase_ts), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int rp_probe_read_handover_ts(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rp::g_handover_ts), sizeof(rp::g_handover_ts), 0, hipMemcpyDeviceToHost);
}
#endif
