"""oracle/ — CPU restatement (plain PyTorch fp32) of the reference's training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under feed_forward_vqgan_clip_amd/ may import this
package; only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and
only as the checker / reported baseline — never as the thing measured or shipped.

Every function is a pure function of (state_dict, inputs) keyed by the REFERENCE's
state_dict names, so the same weights can be fed to the product modules
and to the oracle.  Each function cites the reference file:line it restates.

Pinning status:
  pinned by golden vectors generated from the importable reference (tools/gen_golden.py ->
  tests/golden/*.npz, checked in tests/test_oracle_golden.py):
      mappers.mixer_forward, mappers.vitgan_forward, mappers.simple_vitgan_forward,
      clip.encode_image / clip.encode_text (reference: cloob.CLIP, same architecture as
      clip.model.CLIP), step.clamp_with_grad, step.replace_grad, step.vector_quantize,
      step.synth, step.tv_loss, step.make_cutouts (augs=['R'] path), step.spherical_loss,
      and the composed mini train step (loss + every mapper gradient).
  PARITY UNPINNED (third-party code absent from /root/reference and from this image, restated
  from the published algorithm — SURVEY.md App. A):
      vqgan.decode          taming-transformers-rom1504==0.0.6 (requirements.txt:2)
      mappers.xtransformer_forward   x-transformers==0.19.1 (requirements.txt:20)
      kornia==0.5.10 augmentations (requirements.txt:9) — not restated; parity runs use augs=['R'].
"""
