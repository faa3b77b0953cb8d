"""CPU oracle for the audiotools batched-DSP hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``audiotools_amd/`` imports it, and
the product path raises if its HIP library is missing instead of falling back
to anything here.

Layout
------
``oracle/leaves/``   restatements of the third-party leaves the reference
                     calls but that are absent from this image and from
                     ``/root/reference`` (julius 0.2.7, pyloudnorm 0.1.1,
                     librosa 0.10 ``filters.mel``, torchaudio 2.x
                     ``functional.lfilter`` / ``create_dct``).  All versions
                     are UNPINNED upstream (reference ``setup.py:36-59``).
``oracle/ref_import.py``  imports the UNMODIFIED reference package from
                     ``/root/reference`` with those leaves shimmed into
                     ``sys.modules`` (only possible in the build container;
                     the GPU box has no ``/root/reference``).  Used to make
                     the golden fixtures under ``tests/golden/``.
``oracle/restate.py``  stand-alone restatement (torch-CPU / numpy / scipy) of
                     every hot-path function, citing reference file:line.
                     This is what travels to the GPU box.
``oracle/c/``        plain-C DF-I ``lfilter`` + BS.1770 block energies
                     (mirrors torchaudio's ``cpu_lfilter_core_loop``), built
                     by ``oracle/c/Makefile`` into ``oracle/_build/``.
``oracle/make_golden.py``  regenerates ``tests/golden/*.npz`` from the
                     shim-imported reference.

Parity pinning status (see DESIGN.md "Oracle"):
  * loudness: pinned by the reference's literal targets
    (``tests/core/test_loudness.py``) on synthesised conformance signals and
    by the seeded batch test (``:31-52``).
  * low_pass / high_pass, SplitBands/equalizer, convolve, stft<->istft:
    pinned by the reference's reproducible property tests.
  * mel filter VALUES and resample VALUES: **parity unpinned** -- the
    reference only checks shapes/lengths; our leaves restate the published
    librosa / julius algorithms.
"""
