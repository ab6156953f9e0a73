"""Pieces of bench.py (the driver's contract lives there): the CPU stand-in leg, the roofline accounting, the PMC passes
and the comparison legs.  Measurement code only — nothing here is imported by sgnn_amd/."""
HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3    # v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md)
