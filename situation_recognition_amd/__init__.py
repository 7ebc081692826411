"""situation_recognition_amd -- MI355X (gfx950) native hot path of vFones/situation-recognition.

ResNet backbone -> 6-role GGNN -> verb/noun classifiers, forward + backward, as hand-written HIP
kernels behind a C ABI (include/srhip.h, libsrhip.so), with the reference's Python surface on top
(model.FCGGNN / GGSNN / resnet, imsitu_encoder, imsitu_scorer).  GPU only: there is no CPU fallback.
"""
__version__ = "0.1.0"
