"""CPU oracle for the N-Best-ASR-Transformer fine-tuning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``n-best-asr-transformer_amd/`` (the
product) may import this package.  The only legal importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` - and
there only as the checker / the timed CPU baseline, never as the thing shipped.

What it restates (plain PyTorch fp32 on CPU, autograd for the backward):

* ``oracle.encoder``  - the third-party encoder the reference calls at
  /root/reference/models/model.py:43-45,54-56 (``transformers`` BertModel /
  XLMRobertaModel; pinned ``transformers==2.3.0`` in requirements.txt:5, whose
  source is NOT under /root/reference).  Restated from the published BERT
  algorithm; pinned against the *installed* transformers 5.15.0 by
  ``tests/golden/make_golden.py`` (run in the build container) and the
  committed fixtures under ``tests/golden/``.
* ``oracle.stc``      - /root/reference/models/modules/hierarchical_classifier.py:6-60,
  /root/reference/utils/STC_util.py:4-51, /root/reference/n_best_asr_bert.py:145-215.
* ``oracle.bertadam`` - /root/reference/models/optimization.py:162-171,237-302.
* ``oracle.model``    - /root/reference/models/model.py:11-73.
* ``oracle.step``     - one fine-tuning step, /root/reference/n_best_asr_bert.py:242-280.

Parity status: PINNED against outputs of the reference itself.  The generating
script imports /root/reference (Python, read-only) in the build container and
writes numeric fixtures only; see tests/golden/README.md.  The encoder
arithmetic itself is unpinned *by the reference* (it has no tests and its
pinned transformers version is not installable offline) - documented in
DESIGN.md.
"""
