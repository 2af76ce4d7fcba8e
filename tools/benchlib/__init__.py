"""What bench.py needs besides the contract path (r06: split out of a 1300-line bench.py so that the one number the driver trusts can be reviewed):

  workloads   the bench workloads (fib19.bf, the synthetic nested-counter family), the package loader, committed digests
  roofline    the dominant kernel's roofline from the library's HIP-event records, the committed rocprofv3 summary beside it, the FFT kernels
              against BOTH of their bounds, the in-run sustained-clock probe
  cpu         cpu_baseline: the CPU port (oracle/) timed on the host cores, the SimdBackend-shaped lower bound
  probes      the sweep 2^20..2^26, the Poseidon252 point, proofs in flight through the library's pool (child processes)
  shard       one proof over N GPUs in child processes (the in-process transport probe)
  group       the N > 1 headline: replicas first, then ONE proof over the RCCL shard group, watchdog, extra stages
  launcher    `python3 bench.py --gpus N` without a launcher: rank environments, relay of rank 0's line, host-thread pinning

Nothing here is on the product path."""
