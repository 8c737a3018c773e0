"""CPU oracle for the diffusion-sampling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy fp64 for the
noise schedule, plain torch-CPU fp32 ops for the denoiser and sampler update) of
the algorithm the reference implements in

    diffusion/gaussian_diffusion.py, diffusion/respace.py,
    model/mdm.py, model/mdm_old.py, model/local_attention.py, model/cfg_sampler.py

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / timed CPU baseline.  Nothing under
``gesturediffusion_amd/`` imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this code.

Parity status: PINNED.  The reference ships no tests or golden vectors for this
path, so the oracle is pinned against outputs of the
reference itself, imported in the build container by
``oracle/tools/make_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` re-checks them on every CPU test run).
"""
