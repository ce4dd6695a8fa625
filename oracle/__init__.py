"""CPU oracle for the CellRegMap per-variant score-test path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cellregmap_amd/`` may import,
link or execute anything in this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and only as the checker / the reported CPU baseline.

What is restated here (plain numpy + one small C file, no GPU):

========================  =====================================================
module                    follows
========================  =====================================================
``oracle.sugar``          numpy-sugar >= 1.5.1 (``ddot``, ``economic_svd``,
                          ``economic_qs``, ``economic_qs_linear``, ``epsilon``);
                          in-tree twins at /root/reference cellregmap/_math.py:204-256
``oracle.brent``          brent-search (``bracket`` + ``brent``; Brent 1973
                          ``localmin``) as driven by optimix ``maximize_scalar``
``oracle.lmm``            glimix-core >= 3.1.12 ``LMM`` / ``FastScanner`` as used
                          at cellregmap/_cellregmap.py:254-255,292-309,351-357
``oracle.davies``         chiscore >= 0.2.3 ``davies_pvalue`` / ``liu_sf`` over
                          chi2comb (Davies 1980, AS 155 ``qfc``) -> ``qfc.c``
``oracle.scoretest``      cellregmap/_math.py:33-160 (implicit QS algebra + the
                          dense textbook definitions)
``oracle.crm``            cellregmap/_cellregmap.py:63-131, 137-244 (effect
                          sizes), 246-314, 317-440, 443-469, 471-587, 589-682
========================  =====================================================

PARITY STATUS
-------------
* ``oracle.scoretest`` is PINNED: against the reference's own known-answer
  values (cellregmap/test/test_math.py:55-83) and against golden vectors
  produced by importing the reference's ``_math.py`` in the build container
  (``tests/golden/make_math_golden.py``).
* ``oracle.lmm``, ``oracle.brent``, ``oracle.davies`` restate third-party
  packages whose sources are absent from /root/reference and from this image
  (glimix-core, brent-search, optimix, numpy-sugar, chiscore, chi2comb; no
  network).  The reference holds no golden vector for rho*, v0, v1 or any
  Davies p-value, so for these pieces: **parity unpinned** -- they are written
  from the published algorithms and checked against independent mathematics
  (dense REML likelihood, numerical Imhof integral, scipy distributions) and,
  for ``qfc.c``, against the published table of AS 155 (Davies 1980, Table 1);
  ``oracle.lmm`` also reproduces the worked examples of glimix-core's own
  documentation (lml to 13 digits; values quoted from memory, no network).
  Only the Liu branch is pinned by the reference (test_math.py:76-83).  The
  effect-size functions of ``oracle.crm`` sit on ``oracle.lmm`` and share its
  status; ``compute_maf`` is pinned by the reference's doctest vector
  (_cellregmap.py:600-609).
"""
