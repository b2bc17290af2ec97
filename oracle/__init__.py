"""CPU oracle for the tauLDR / SDDM hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a CPU restatement (numpy / torch-CPU fp32) of the reference algorithms on the
hot path named in BASELINE.json (forward CTMC q_{t|0}, reverse rates, tau-leaping / Euler / PC
sampling steps, noising, losses, score networks).  Every function cites the reference file:line
it restates.  It is pinned against outputs of the reference itself (`oracle/gen_golden.py`
imports /root/reference in the build container and freezes `tests/golden/*.npz`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package -- as the checker, never as the thing measured or shipped.  The product path
(`continuous-time-diffusion-models-for-discrete-data_amd/`) never imports it and fails loudly
when the HIP library is missing.
"""
