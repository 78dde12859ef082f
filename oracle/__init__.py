"""CPU oracle for the Melspectrogram hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  The product
(``torchaudio-contrib_amd/``) never imports this package and fails loudly when its
HIP library is missing.

Two independent restatements of the reference algorithm live here:

* ``torch_ref``  – stock ``torch`` CPU ops in the reference's own op order
  (``torchaudio_contrib/functional.py``), i.e. what the reference actually executes
  on a CPU once ``torch.stft`` is told ``return_complex=True``.  This is the timed
  ``cpu_baseline`` ("port") and the primary parity oracle.
* ``numpy_ref``  – float64 numpy framing + ``rfft``; shares no code with torch's FFT,
  used to cross-check ``torch_ref`` and the HIP kernels.

Pinning: both are checked against golden vectors captured from the *unmodified*
reference imported from ``/root/reference`` (``tests/golden/make_golden.py``, fixtures in
``tests/golden/``) by ``tests/test_oracle_golden.py``.
"""
