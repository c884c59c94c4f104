"""Per-launch-group budget of the 1024^2 step: four throughput floors at the sustained clock -
matrix pipe, LDS, VALU issue, HBM - and the ENERGY floor the power-limited part actually obeys, from the measured price list
(profiles/r05_energy_prices.txt) and per-group instruction counts.  Counts per group are entered by hand below from the kernels'
structure and ISA (scripts/isa_loops.py, DESIGN_LOG 10.4's table); times from the round's same-box slot profile.

    python scripts/energy_budget.py            -> markdown table on stdout"""
B = 128
F_SUST = 1.9e9          # shader clock the step sustains (bench key sustained_sclk_mhz: 1.89-1.98 GHz)
P_CAP, P_FIXED = 1290.0, 270.0   # W: socket power under the step (r03 power samples, r05 ubench); what the chip draws with clocks up and nothing issuing
E_MFMA, E_LDS, E_VALU, E_DMA_KB, E_HBM_B = 11.5e-9, 4.1e-9, 1.1e-9, 20.4e-9, 0.11e-9   # J per wave-instruction / KB / byte (VALU: plain 0.86, packed 1.87, mixed)
SIMDS, CUS = 1024, 256

# name: (ms per step, executed GMAC per frame, ds_read_b128 per MFMA, VALU wave-instr per MFMA, LDS-direct KB per MFMA, HBM GB per step, extra VALU wave-instr (G) without MFMAs)
G = {
    "1024^2 block: fused walk":                 (6.40, 20.3 * 1.0625, 1.13, 5.3, 0.03, 6.86, 0.0),
    "conv1 32^2 / 64^2 / 128^2 (dma 256-ch)":   (0.55 + 1.92 + 2.37, 21.72, 0.75, 2.3, 0.145, 5.43, 0.0),
    "up 32->64, 64->128, 128->256 (tconv + FIR pairs)": (0.74 + 0.22 + 1.41 + 0.43 + 1.43 + 0.86, 12.08, 0.83, 1.9, 0.257, 15.5, 0.176),
    "up 256->512 (tconv_fir)":                  (3.31, 4.83 * 1.42, 1.15, 4.7, 0.257, 7.70, 0.0),
    "conv1 512^2 (hires)":                      (3.05, 9.66, 1.3, 4.6, 0.0, 11.39, 0.0),
    "conv1 256^2 (dma 128-ch)":                 (2.49, 9.66, 1.0, 2.8, 0.163, 4.54, 0.0),
    # (executed MACs: these layers run in phase form - the FIR folded into the weights, 4x the minimal MACs of their up-layers)
    "<= 16^2 layers, toRGB <= 128^2, styles":   (2.2, 4.0, 2.0, 6.0, 0.0, 1.5, 0.0),
    "noise maps":                               (0.47, 0.0, 0.0, 0.0, 0.0, 1.39, 0.17),
}

print("| launch group | ms | MFMA floor | LDS floor | VALU floor | HBM floor | energy floor | measured / tightest | energy: MFMA / LDS+DMA / VALU / HBM / fixed (J) |")
print("|---|---|---|---|---|---|---|---|---|")
tot_t = tot_e = 0.0
for name, (ms, gmac, rd, va, dma, hbm, xv) in G.items():
    n_mfma = gmac * 1e9 * B / 16384
    n_rd, n_va, kb = n_mfma * rd, n_mfma * va + xv * 1e9, n_mfma * dma
    f_mfma = n_mfma * 32 / SIMDS / F_SUST * 1e3
    f_lds = (n_rd + kb) * 1024 / (CUS * 256) / F_SUST * 1e3           # 256 B / clk / CU (ds_read_b128; the fills write at least as slowly)
    f_valu = n_va * 4 / SIMDS / F_SUST * 1e3                          # 4 cycles per wave-instruction and SIMD
    f_hbm = hbm * 1e9 / 6.3e12 * 1e3
    e = (n_mfma * E_MFMA, n_rd * E_LDS + kb * E_DMA_KB, n_va * E_VALU, hbm * 1e9 * E_HBM_B)
    f_en = sum(e) / (P_CAP - P_FIXED) * 1e3
    tight = max(f_mfma, f_lds, f_valu, f_hbm, f_en)
    tot_t += ms
    tot_e += sum(e) + P_FIXED * ms * 1e-3
    print(f"| {name} | {ms:.2f} | {f_mfma:.2f} | {f_lds:.2f} | {f_valu:.2f} | {f_hbm:.2f} | **{f_en:.2f}** | {ms / tight:.2f} | "
          f"{e[0]:.2f} / {e[1]:.2f} / {e[2]:.2f} / {e[3]:.2f} / {P_FIXED * ms * 1e-3:.2f} |")
print(f"| step | {tot_t:.1f} | | | | | | | accounted {tot_e:.1f} J of {P_CAP * tot_t * 1e-3:.1f} J measured (power x time) |")
