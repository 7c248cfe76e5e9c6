

def test_tsv_rows_are_printf_16_significant_digits(oa, tmp_path):
    """layout.cpp:10-35: `idx X Y component`, two rows per node, rows grouped by component, numbers as the
    reference's stream with precision 16 prints them (= printf %.16g), also for awkward values."""
    import ctypes as C
    from odgi_amd._lib import lib
    n = 70_000                                              # more than one formatting chunk
    rs = np.random.RandomState(2)
    X = rs.rand(2 * n) * 3e7
    Y = (rs.rand(2 * n) - 0.5) * 1e4
    X[:12] = [0.0, 1.0, -1.0, 1e-300, 1e300, 123456789012345678.0, 0.1, 1.0 / 3.0, 2.5e-5, 1000.0, 1e16, 123456.7890123456789]
    comp = (np.arange(n) % 3).astype(np.uint32)
    f = tmp_path / "o.tsv"
    assert lib.pgsgd_write_tsv(str(f).encode(), n, comp.ctypes.data_as(C.POINTER(C.c_uint32)), 3, X.ctypes.data_as(C.POINTER(C.c_double)),
                               Y.ctypes.data_as(C.POINTER(C.c_double))) == 0
    lines = f.read_text().split("\n")
    assert lines[0] == "idx\tX\tY\tcomponent" and lines[-1] == "" and len(lines) == 2 * n + 2
    order = np.concatenate([np.where(comp == c)[0] for c in range(3)])
    want = []
    for i in order:
        for e in (2 * i, 2 * i + 1):
            want.append("%d\t%.16g\t%.16g\t%d" % (e, X[e], Y[e], comp[i]))
    assert lines[1:-1] == want
