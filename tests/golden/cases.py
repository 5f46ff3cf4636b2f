"""Golden-vector case list shared by make_golden.py (build container) and the parity tests.

Field meanings follow the reference CLI (src/sdrterm.py:54-107): ``code`` is the --encoding name
(misc/file_util.py:44-56), ``q`` is --decimation, ``center`` is --center-frequency, ``vfos`` is the
--vfos csv (rows = csv entries + center, then center last, vfo_processor.py:43-47), ``omega_out`` is
--omega-out, ``normalize`` / ``correct_iq`` are --normalize-input / --correct-iq, ``smooth`` is --smooth-output.
"""
READ_SIZE = 131072  # bytes per chunk, misc/read_file.py:38

_VFOS8 = '-200000,-150000,-100000,-50000,50000,100000,150000,200000'
_VFOS7_BENCH = '-9000000,-6000000,-3000000,3000000,6000000,9000000,12000000'     # bench.py simo_rows(8) less the centre


def _c(name, code, q, mode, center=0, vfos=None, omega_out=5000, fs=1024000, n_chunks=2, seed=1, amp=3000,
       **kw):
    d = dict(name=name, code=code, q=q, mode=mode, center=center, vfos=vfos, omega_out=omega_out, fs=fs,
             n_chunks=n_chunks, seed=seed, amp=amp)
    d.update(kw)
    return d


CASES = [
    _c('h_q64_fm_c0', 'h', 64, 'fm', seed=11),
    _c('h_q64_fm_c', 'h', 64, 'fm', center=15700, seed=12),
    _c('h_q64_am_norm_c', 'h', 64, 'am', center=-15700, normalize=True, seed=13),
    _c('h_q64_am_raw', 'h', 64, 'am', center=15700, seed=14),
    _c('h_q32_fm_c', 'h', 32, 'fm', center=20300, seed=15),
    _c('h_q32_am_norm_c0', 'h', 32, 'am', normalize=True, seed=16),
    _c('f_q64_fm_c', 'f', 64, 'fm', center=15700, seed=17),
    _c('f_q32_am_c0', 'f', 32, 'am', seed=18),
    _c('h_q64_fm_vfo9', 'h', 64, 'fm', center=15700, vfos=_VFOS8, seed=19),
    _c('h_q32_am_norm_vfo9_iq', 'h', 32, 'am', center=15700, vfos=_VFOS8, normalize=True, correct_iq=True,
       n_chunks=3, dc=(500, -300), seed=20),
    _c('h_q32_fm_vfo9_iq', 'h', 32, 'fm', center=0, vfos=_VFOS8, correct_iq=True, n_chunks=3, dc=(500, -300),
       seed=21),
    _c('h_q64_fm_iq', 'h', 64, 'fm', center=15700, correct_iq=True, n_chunks=3, dc=(500, -300), seed=22),
    _c('B_q64_fm_c', 'B', 64, 'fm', center=15700, amp=100, seed=23),
    _c('h_be_q64_fm_c', 'h', 64, 'fm', center=15700, big_endian=True, seed=24),
    _c('d_q64_fm_c', 'd', 64, 'fm', center=15700, seed=25),
    _c('i_q16_fm_c', 'i', 16, 'fm', center=15700, amp=100000, seed=26),
    _c('b_q128_fm_c0', 'b', 128, 'fm', omega_out=2500, amp=100, seed=27),
    _c('H_q64_am_norm_c', 'H', 64, 'am', center=15700, normalize=True, seed=28),
    _c('h_q256_fm_c', 'h', 256, 'fm', center=15700, omega_out=1500, seed=29),
    # --smooth-output (Savitzky-Golay on the written samples, dsp_processor.py:159-160): odd and even window lengths
    _c('h_q64_fm_c_smooth11', 'h', 64, 'fm', center=15700, smooth=11, seed=30),
    _c('h_q32_am_norm_c0_smooth8', 'h', 32, 'am', normalize=True, smooth=8, seed=31),
    # small decimation factors (the CLI default is --decimation 2, src/sdrterm.py:80) and many outputs per chunk
    _c('h_q2_fm_c', 'h', 2, 'fm', center=15700, omega_out=100000, seed=32),
    _c('h_q4_am_c0', 'h', 4, 'am', omega_out=60000, seed=33),
    _c('h_q2_am_norm_c', 'h', 2, 'am', center=-15700, normalize=True, omega_out=100000, seed=34),
    _c('h_q8_fm_c', 'h', 8, 'fm', center=15700, omega_out=30000, seed=35),
    _c('B_q16_fm_c', 'B', 16, 'fm', center=15700, omega_out=15000, amp=100, seed=36),
    _c('f_q4_fm_c', 'f', 4, 'fm', center=15700, omega_out=60000, seed=37),
    # BASELINE.json configs at their own parameters (round 3): configs[1] (256 MS/s, centre 15 kHz, omega_out 12.5 kHz),
    # configs[2] (7 csv rows + centre = 8 rows as _nFreq counts them, q = 32, --correct-iq, one run per demodulation),
    # configs[4] (1 GS/s float32, rows from both ends and the middle of bench.py's 256-row layout)
    _c('h_q64_fm_bench', 'h', 64, 'fm', center=15000, omega_out=12500, fs=256000000, seed=38),
    _c('h_q32_fm_vfo8_iq_bench', 'h', 32, 'fm', center=15000, vfos=_VFOS7_BENCH, omega_out=12500, fs=256000000,
       correct_iq=True, n_chunks=3, dc=(500, -300), seed=39),
    _c('h_q32_am_norm_vfo8_iq_bench', 'h', 32, 'am', center=15000, vfos=_VFOS7_BENCH, omega_out=12500, fs=256000000,
       normalize=True, correct_iq=True, n_chunks=3, dc=(500, -300), seed=40),
    _c('f_q64_fm_vfo5_1g', 'f', 64, 'fm', center=15000, vfos='-381000000,-3000000,6000000,384000000', omega_out=12500,
       fs=1000000000, seed=41),
    # the same normalise flag in an FM and an AM run of one 9-row layout: what a context with per-row modes must equal
    _c('h_q32_fm_norm_vfo9_iq', 'h', 32, 'fm', center=15700, vfos=_VFOS8, normalize=True, correct_iq=True,
       n_chunks=3, dc=(500, -300), seed=20),
    # ---- round 4: classes of input other than moderate white noise (prng.case_ints documents the ``sig`` kinds) ----------------
    # (i) the encodings' rails and the byte-slice corner values (x = 256 hi + lo' + 128), in every position incl. the 28-sample edges
    _c('h_q64_fm_rails_c0', 'h', 64, 'fm', n_chunks=3, amp=8000, seed=50, sig=dict(kind='rails', chunks=['corners', 'const', 'edges'])),
    _c('h_q64_fm_rails_c', 'h', 64, 'fm', center=15700, n_chunks=3, amp=8000, seed=51, sig=dict(kind='rails', chunks=['corners', 'two', 'edges'])),
    _c('h_q32_am_rails_c', 'h', 32, 'am', center=15700, n_chunks=3, amp=8000, seed=52, sig=dict(kind='rails', chunks=['two', 'corners', 'const'])),
    _c('B_q64_fm_rails_c', 'B', 64, 'fm', center=15700, n_chunks=3, amp=100, seed=53, sig=dict(kind='rails', chunks=['corners', 'two', 'edges'])),
    _c('B_q64_fm_rails_c0', 'B', 64, 'fm', n_chunks=3, amp=100, seed=54, sig=dict(kind='rails', chunks=['two', 'const', 'corners'])),
    _c('H_q64_fm_rails_c', 'H', 64, 'fm', center=15700, n_chunks=3, amp=8000, seed=55, sig=dict(kind='rails', chunks=['corners', 'two', 'edges'])),
    _c('b_q32_fm_rails_c', 'b', 32, 'fm', center=15700, n_chunks=2, amp=100, seed=56, sig=dict(kind='rails', chunks=['two', 'corners'])),
    _c('h_q64_fm_rails_iq', 'h', 64, 'fm', center=15700, correct_iq=True, n_chunks=3, amp=8000, seed=57,
       sig=dict(kind='rails', chunks=['corners', 'two', 'edges'])),
    _c('h_q16_fm_rails_c', 'h', 16, 'fm', center=15700, omega_out=15000, n_chunks=2, amp=8000, seed=58, sig=dict(kind='rails', chunks=['two', 'edges'])),
    _c('h_q2_fm_rails_c', 'h', 2, 'fm', center=15700, omega_out=100000, n_chunks=2, amp=8000, seed=59, sig=dict(kind='rails', chunks=['corners', 'edges'])),
    # (ii) the class of stream bench.py times: an FM tone (1 kHz modulation, 2.5 kHz deviation) of amplitude 8000 + bell noise sigma 300
    _c('h_q64_fm_tone_bench', 'h', 64, 'fm', center=15000, omega_out=12500, fs=256000000, seed=60,
       sig=dict(kind='tone', a=8000, f=15000, fm=1000, dev=2500, noise=260)),
    _c('h_q64_fm_tone_1m', 'h', 64, 'fm', center=15700, seed=61, sig=dict(kind='tone', a=8000, f=15700, fm=1000, dev=2500, noise=260)),
    _c('B_q64_fm_tone_1m', 'B', 64, 'fm', center=15700, seed=62, amp=100, sig=dict(kind='tone', a=62, f=15700, fm=1000, dev=2500, noise=2)),
    _c('h_q64_am_tone_norm', 'h', 64, 'am', center=15700, normalize=True, seed=63, sig=dict(kind='tone', a=8000, f=15700, fm=1000, dev=2500, noise=260)),
    _c('h_q64_fm_tone_iq_dc', 'h', 64, 'fm', center=15700, correct_iq=True, n_chunks=3, dc=(20000, -20000), seed=64,
       sig=dict(kind='tone', a=8000, f=15700, fm=1000, dev=2500, noise=260)),
    _c('h_q32_fm_tone_vfo8_iq_dc_bench', 'h', 32, 'fm', center=15000, vfos=_VFOS7_BENCH, omega_out=12500, fs=256000000, correct_iq=True,
       n_chunks=3, dc=(20000, -20000), seed=65, sig=dict(kind='tone', a=8000, f=15000, fm=1000, dev=2500, noise=260)),
    # (iii) a weak in-band tone under a full-scale out-of-band one (the worst case of the int8 path's 2^-39 weight image); the row sits
    # on 15625 Hz = 1000 fs/65536, the weak tone 1 kHz above it, the strong one 200 kHz (far) or 12.5 kHz (transition band) above it
    _c('h_q64_fm_blk60', 'h', 64, 'fm', center=15625, seed=66, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-60)),
    _c('h_q64_fm_blk80', 'h', 64, 'fm', center=15625, seed=67, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-80)),
    _c('h_q64_fm_blk80_near', 'h', 64, 'fm', center=15625, seed=68, sig=dict(kind='blocker', a1=32000, f1=28125, f2=16625, db=-80)),
    _c('h_q64_am_blk60', 'h', 64, 'am', center=15625, seed=69, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-60)),
    _c('h_q32_fm_blk60_iq_dc', 'h', 32, 'fm', center=15625, correct_iq=True, n_chunks=3, dc=(20000, -20000), seed=70,
       sig=dict(kind='blocker', a1=12000, f1=215625, f2=16625, db=-60)),
    _c('B_q64_fm_blk40', 'B', 64, 'fm', center=15625, amp=100, seed=71, sig=dict(kind='blocker', a1=120, f1=215625, f2=16625, db=-40)),
    _c('h_q64_fm_blk80_c0', 'h', 64, 'fm', center=0, seed=72, sig=dict(kind='blocker', a1=32000, f1=200000, f2=1000, db=-80)),
    # (iv) full-scale impulses on the first / last samples of a chunk (scipy's odd extension by 27, dsp_processor.py:147) over noise
    _c('h_q64_fm_impulse_c', 'h', 64, 'fm', center=15700, amp=1000, seed=73, sig=dict(kind='impulse', at=[0, 26, -27, -1])),
    _c('h_q64_fm_impulse_c0', 'h', 64, 'fm', amp=1000, seed=74, sig=dict(kind='impulse', at=[0, 26, -27, -1])),
    _c('h_q16_fm_impulse_c', 'h', 16, 'fm', center=15700, omega_out=15000, amp=1000, seed=75, sig=dict(kind='impulse', at=[0, 26, -27, -1])),
    _c('h_q2_am_impulse_c', 'h', 2, 'am', center=15700, omega_out=100000, amp=1000, seed=76, sig=dict(kind='impulse', at=[0, 26, -27, -1])),
    _c('B_q64_fm_impulse_c', 'B', 64, 'fm', center=15700, amp=30, seed=77, sig=dict(kind='impulse', at=[0, 26, -27, -1])),
    # (v) all-zero chunks between (and before) loud ones: the reference writes exact zeros there (angle(0) = 0)
    _c('h_q64_fm_zero_c', 'h', 64, 'fm', center=15700, n_chunks=3, amp=8000, seed=78, sig=dict(kind='zero', chunks=[1])),
    _c('h_q64_fm_zero_c0', 'h', 64, 'fm', n_chunks=3, amp=8000, seed=79, sig=dict(kind='zero', chunks=[0, 2])),
    _c('h_q32_am_zero_vfo9', 'h', 32, 'am', center=15700, vfos=_VFOS8, n_chunks=3, amp=8000, seed=80, sig=dict(kind='zero', chunks=[1])),
    _c('h_q32_fm_zero_vfo9', 'h', 32, 'fm', center=15700, vfos=_VFOS8, n_chunks=3, amp=8000, seed=81, sig=dict(kind='zero', chunks=[1])),
    _c('B_q64_fm_zero_c', 'B', 64, 'fm', center=15700, n_chunks=3, amp=100, seed=82, sig=dict(kind='zero', chunks=[1])),
    _c('H_q64_fm_zero_c', 'H', 64, 'fm', center=15700, n_chunks=3, amp=8000, seed=83, sig=dict(kind='zero', chunks=[1])),
    # (--correct-iq: the silent chunk is -offset, a slowly decaying exponential; on the row's own frequency it is the signal.  With the
    # row 15.7 kHz away it would be 164 dB down in the stop band -- an FM angle of 1e-13 of the input peak, which no arithmetic
    # other than the reference's own operation order reproduces to 1e-6: DESIGN section 1, "conditioning")
    _c('h_q64_fm_zero_iq_dc_c0', 'h', 64, 'fm', center=0, correct_iq=True, n_chunks=3, dc=(20000, -20000), amp=8000, seed=84,
       sig=dict(kind='zero', chunks=[1])),
    _c('f_q64_fm_zero_c', 'f', 64, 'fm', center=15700, n_chunks=3, amp=8000, seed=85, sig=dict(kind='zero', chunks=[1])),
    _c('h_q4_fm_zero_c', 'h', 4, 'fm', center=15700, omega_out=60000, n_chunks=3, amp=8000, seed=86, sig=dict(kind='zero', chunks=[1])),
    _c('h_q16_fm_zero_c', 'h', 16, 'fm', center=15700, omega_out=15000, n_chunks=3, amp=8000, seed=87, sig=dict(kind='zero', chunks=[1])),
    # a stream that STARTS silent under --correct-iq (offset exactly zero): the reference writes zeros for that chunk
    _c('h_q64_fm_zero_iq_start', 'h', 64, 'fm', center=15700, correct_iq=True, n_chunks=3, amp=8000, seed=88, sig=dict(kind='zero', chunks=[0])),
    # ---- round 5 ----------------------------------------------------------------------------------------------------------------
    # --normalize-input at the decimations the wide kernel takes in several passes (signed 8-bit records have no slice
    # offset, their normalisation response was not split over the passes)
    _c('b_q16_am_norm_c', 'b', 16, 'am', center=15700, normalize=True, omega_out=15000, amp=100, seed=89),
    _c('b_q16_fm_norm_c0', 'b', 16, 'fm', normalize=True, omega_out=15000, amp=100, seed=90),
    _c('B_q16_fm_norm_c', 'B', 16, 'fm', center=15700, normalize=True, omega_out=15000, amp=100, seed=91),
    _c('h_q16_fm_norm_c', 'h', 16, 'fm', center=15700, normalize=True, omega_out=15000, seed=92),
    _c('h_q8_am_norm_c0', 'h', 8, 'am', normalize=True, omega_out=30000, seed=93),
    _c('h_q4_fm_norm_c', 'h', 4, 'fm', center=15700, normalize=True, omega_out=60000, seed=94),
    # the ill-conditioned class, kept as a stated limit (tests: xfail(strict) with the measured bound): a silent chunk under
    # --correct-iq after a DC of 20000 seen from a row 15.7 kHz away -- the chunk is the decaying offset, 164 dB down the stop band
    # at this row, and the reference's FM output is the angle of its own rounding at that level (read_file.py:64-77 ->
    # demodulation.py:30-32)
    _c('h_q64_fm_zero_iq_dc_c', 'h', 64, 'fm', center=15700, correct_iq=True, n_chunks=3, dc=(20000, -20000), amp=8000, seed=84,
       sig=dict(kind='zero', chunks=[1])),
    # in-band content 100 / 110 / 120 dB under a full-scale out-of-band tone: at 16 bits that is a tenth to a thirtieth of an LSB,
    # i.e. the quantisation residue itself -- where the 40-bit operator image of the int8 forms ends and float64 carries on
    _c('h_q64_fm_blk100', 'h', 64, 'fm', center=15625, seed=95, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-100)),
    _c('h_q64_fm_blk110', 'h', 64, 'fm', center=15625, seed=96, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-110)),
    _c('h_q64_fm_blk120', 'h', 64, 'fm', center=15625, seed=97, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-120)),
    # the same class at 1024 outputs per chunk (the sixteen-lanes-per-item demodulation stage) and with two rows of which only one
    # is weak (the other sits ON the strong tone): the second look of stage_a = auto is per (chunk, row)
    _c('h_q32_fm_blk120', 'h', 32, 'fm', center=15625, seed=98, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-120)),
    _c('h_q64_fm_blk120_vfo2', 'h', 64, 'fm', center=15625, vfos='200000', seed=99, sig=dict(kind='blocker', a1=32000, f1=215625, f2=16625, db=-120)),
]

BY_NAME = {c['name']: c for c in CASES}

# the plots' FFT taps (make_golden.py: plot_taps): encoding, --normalize-input; one reader chunk each
TAP_CASES = [('h', False), ('B', True), ('f', False)]
TAP_FS, TAP_OFFSET, TAP_SEED = 1024000, 15700, 5
