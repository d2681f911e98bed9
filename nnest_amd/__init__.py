"""nnest_amd: MI355X-native implementation of the nnest flow-transform + batched-proposal + likelihood
+ flow-training hot path (reference: adammoss/nnest v0.4.2).  See DESIGN.md / INTEGRATION.md."""
__version__ = '0.1.0'
