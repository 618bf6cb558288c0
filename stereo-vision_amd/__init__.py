"""MI355X-native diffusion_stereo hot path: host-side Python surface.

The product is the C-ABI library built from csrc/ (include/diffstereo.h); this
package only binds it for the tests and the benchmark (`diffstereo`), generates
synthetic inputs (`synth`) and holds the row-band multi-GPU host logic (`bands`) and the
reference's `main` over it (`pipeline`).
The directory name is not a Python identifier: load it with
`__graft_entry__.load_package()`, which registers it as `stereo_vision_amd`.
"""
