"""
Input parsing (boundary producer of the hot path), pinned by the reference's own parse tests
(/root/reference/tests/test_inbred.py:9-18) on the reference's sample files (tests/conftest.py:27-35;
copied as data fixtures: tests/golden/701_501.filter.vcf.gz, 701_502.filter.bed).
"""
import gzip
import json
import os
import shutil

import numpy as np
import pytest

from snpmatch_amd.core import _vcf, parsers


def test_vcf_parse_reference_pins(golden_dir, tmp_path):
    src = os.path.join(golden_dir, "701_501.filter.vcf.gz")
    vcf = tmp_path / "701_501.filter.vcf"
    with gzip.open(src, "rb") as fi, open(vcf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    snps = parsers.ParseInputs(inFile=str(vcf), logDebug=True)
    assert len(snps.chrs) == 7545                       # test_inbred.py:10
    assert snps.chrs[0] == 'Chr1'                       # :11
    assert snps.gt[0] == '0/0'                          # :12
    vals, counts = np.unique(snps.gt, return_counts=True)
    assert dict(zip(vals.tolist(), counts.tolist())) == {'0/0': 6579, '0/1': 110, '1/1': 856}
    assert snps.pos[0] == 13226
    # PL 0,9,87 -> exp(-PL/10) (core/parsers.py:147-150); depth from INFO/DP
    np.testing.assert_array_equal(snps.wei[0], np.exp(np.array([0.0, 9.0, 87.0]) / (-10)))
    assert snps.wei.shape == (7545, 3) and snps.dp[0] == 3
    assert np.all((snps.wei > 0) & (snps.wei <= 1))
    # cache + stats files, as the reference writes them (core/parsers.py:85-86, 96-116); the cache is written by a background thread
    snps.wait_for_cache()
    assert os.path.isfile(str(vcf) + ".snpmatch.npz")
    stats = json.load(open(str(vcf) + ".snpmatch.stats.json"))
    assert stats["num_of_snps"] == 7545 and stats["interpretation"]["case"] == 0
    assert stats["percent_heterozygosity"] == 110 / 7545.0
    again = parsers.ParseInputs(inFile=str(vcf), logDebug=False)          # served from the .npz cache
    assert np.array_equal(again.pos, snps.pos) and np.array_equal(again.wei, snps.wei)
    assert np.array_equal(again.gt, snps.gt) and np.array_equal(again.chrs, snps.chrs)
    # gz input goes through the same reader
    gz = tmp_path / "copy.vcf.gz"
    shutil.copyfile(src, gz)
    z = parsers.ParseInputs(inFile=str(gz), logDebug=False)
    assert len(z.chrs) == 7545 and np.array_equal(z.wei, snps.wei)


def test_bed_parse_reference_pins(golden_dir, tmp_path):
    bed = tmp_path / "701_502.filter.bed"
    shutil.copyfile(os.path.join(golden_dir, "701_502.filter.bed"), bed)
    snps = parsers.ParseInputs(inFile=str(bed), logDebug=True)
    assert len(snps.chrs) == 10000                      # test_inbred.py:15
    assert snps.chrs[0] == '1'                          # :16
    assert snps.gt[0] == '0/0'                          # :17
    assert snps.pos[1] == 51103                         # :18
    # hard weights: one-hot on the called genotype (core/parsers.py:126-129)
    assert set(np.unique(snps.wei).tolist()) <= {0.0, 1.0}
    assert np.all(snps.wei.sum(axis=1) <= 1.0)
    codes = parsers.parseGT(snps.gt)
    assert np.all(snps.wei[codes == 0, 0] == 1) and np.all(snps.wei[codes == 1, 2] == 1)
    # the reference crashes on BED depth "NA" (np.nanmean("NA")); here the stats file reports NaN
    stats = json.load(open(str(bed) + ".snpmatch.stats.json"))
    assert stats["num_of_snps"] == 10000 and np.isnan(stats["depth"])
    # hard calls carry the two-entry dictionary form of their weights (Genotyper's one-call path uploads it instead of the fp64 triples)
    pair = snps.weight_codes()
    assert pair is not None and pair[1].tolist() == [0.0, 1.0] and pair[0].dtype == np.uint16
    assert np.array_equal(pair[1][pair[0]].view(np.uint64), snps.wei.view(np.uint64))
    assert parsers._hard_call_codes(np.array([[1.0, 0.5, 0.0]])) is None and parsers._hard_call_codes(np.array([[1.0, -0.0, 0.0]])) is None


def test_parse_gt_and_chr_names():
    gt = np.array(["0/0", "1/1", "0/1", "1/0", "./.", "1/2"])
    assert parsers.parseGT(gt).tolist() == [0, 1, 2, 2, -1, 0]
    assert parsers.parseGT(np.array(["0|0", "1|1", "0|1", ".|."])).tolist() == [0, 1, 2, -1]
    assert parsers.parseGT(np.array(["0", "1", "2"])).tolist() == [0, 1, 2]
    assert parsers.parseGT(np.array([])).tolist() == []
    assert parsers.snp_binary_to_gt([-1, 0, 1, 2]).tolist() == [b"./.", b"0/0", b"1/1", b"0/1"]
    ins = parsers.ParseInputs("")
    ins.load_snp_info(["Chr2", "chr2", "CHR1", "1", "M"], [1, 2, 3, 4, 5], "", np.nan, 0)
    ins.filter_chr_names()
    assert ins.g_chrs.tolist() == ["2", "2", "1", "1", "M"]
    assert ins.g_chrs_ids.tolist() == ["2", "1", "M"]           # order of first appearance
    assert ins.g_chr_codes.tolist() == [0, 0, 1, 1, 2]
    # unsorted input (a chromosome in several runs) and the plain per-string route give the same three arrays
    rng = np.random.default_rng(8)
    names = rng.choice(np.array(["Chr1", "chr1", "2", "CHR2", "Pt", "chrM", "3"]), size=500)
    ins.load_snp_info(names, np.arange(500), "", np.nan, 0)
    ins.filter_chr_names()
    import re
    bare = np.array([re.sub("chr", "", c, flags=re.IGNORECASE) for c in names.tolist()])
    assert np.array_equal(ins.g_chrs, bare)
    first_seen = []
    for c in bare.tolist():
        if c not in first_seen:
            first_seen.append(c)
    assert ins.g_chrs_ids.tolist() == first_seen
    assert np.array_equal(ins.g_chrs_ids[ins.g_chr_codes], bare)
    # the integer route of parseGT (three-character calls) equals the string route on every call text that occurs
    calls = np.array(["0/0", "1/1", "0/1", "1/0", "./.", "1/2", "2/1", "0/.", "./1", "0|1", "1|1"])
    mixed = rng.choice(calls[:9], size=300)
    mixed[0] = "0/1"
    slow = np.zeros(300, dtype=np.int8)
    for pat, code in (("1/1", 1), ("0/1", 2), ("1/0", 2), ("./.", -1)):
        slow[mixed == pat] = code
    assert np.array_equal(parsers.parseGT(mixed), slow)
    assert parsers.parseGT(np.array(["0/1", "10/1", "1/1"])).tolist() == [2, 0, 1]      # wider strings: the string route


def _same_calls(a, b):
    for k in ("samples", "chr", "pos", "gt", "pl", "dp"):
        if b[k] is None:
            assert a[k] is None, k
        else:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    assert a["has_gt"] == b["has_gt"]


def test_native_vcf_reader_equals_python_reader(golden_dir, tmp_path):
    """snpm_vcf_parse (C++, one pass) against the Python reader it accelerates: identical arrays"""
    from snpmatch_amd.core import _vcf
    src = os.path.join(golden_dir, "701_501.filter.vcf.gz")
    _same_calls(_vcf.read_calls(src, native=True), _vcf.read_calls(src, native=False))
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n"
    body = [
        "Chr1\t10\t.\tA\tT\t50\tPASS\tDP=7;AF=0.5\tGT:PL\t0/1:10,0,30\t1/1:99,9,0",
        "Chr1\t20\t.\tA\tT\t50\tPASS\tAF=0.5;DP=12\tGT:DP:PL\t1|1:5:255,30.5,0\t./.:.:.",
        "Chr1\t30\t.\tA\tT\t50\tPASS\t.\tGT\t.\t0/0",                         # bare '.', no PL
        "chr2\t5\t.\tA\tT,G\t50\tPASS\tDP=\tGT:PL\t0/2:1,2,3,4,5,6\t0/0:0,1",   # empty DP, six PL values, short PL
        "chr2\t7\t.\tA\tT\t50\tPASS\tXDP=3\tPL:GT\t.,3,.:0/0\t1e1,+2,.5:1/1",   # PL before GT, '.' items, float forms
        "short\t9\t.\tA",                                                         # short record: skipped
        "",                                                                       # empty line: skipped
        "Mt\t11\t.\tA\tT\t50\tPASS\tDP=3\tGT:PL",                                 # FORMAT but no sample column
        "Mt\t12\t.\tA\tT\t50\tPASS\tDP=4",                                        # eight columns only
    ]
    vcf = tmp_path / "edge.vcf"
    vcf.write_text(head + "\n".join(body) + "\n")
    for sample in (0, 1):
        fast = _vcf.read_calls(str(vcf), samples=(sample,), native=True)
        slow = _vcf.read_calls(str(vcf), samples=(sample,), native=False)
        _same_calls(fast, slow)
    assert fast["gt"][:, 0].tolist() == ["1/1", "./.", "0/0", "0/0", "1/1", "./.", "./."]
    assert fast["pl"][4, 0].tolist() == [10.0, 2.0, 0.5] and fast["dp"].tolist() == [7, 12, -1, -1, -1, 3, 4]
    # a file without PL / DP anywhere
    plain = tmp_path / "plain.vcf"
    plain.write_text(head + "1\t100\t.\tA\tT\t.\t.\t.\tGT\t0/0\t1/1\n1\t200\t.\tA\tT\t.\t.\t.\tGT\t1/1\t0/0\n")
    fast, slow = _vcf.read_calls(str(plain), native=True), _vcf.read_calls(str(plain), native=False)
    _same_calls(fast, slow)
    assert fast["pl"] is None and fast["dp"] is None
    # things the native reader leaves to the generic one
    from snpmatch_amd import _lib
    crlf = tmp_path / "crlf.vcf"
    crlf.write_bytes((head + body[0] + "\n").replace("\n", "\r\n").encode())
    assert _lib.vcf_parse(str(crlf), 0) is None
    odd = tmp_path / "odd.vcf"
    odd.write_text(head + "1\t1_0\t.\tA\tT\t.\t.\t.\tGT\t0/0\t1/1\n")
    assert _lib.vcf_parse(str(odd), 0) is None and _vcf.read_calls(str(odd))["pos"].tolist() == [10]
    odd.write_text(head + "1\t10\t.\tA\tT\t.\t.\tDP= 5\tGT\t0/0\t1/1\n")
    assert _lib.vcf_parse(str(odd), 0) is None and _vcf.read_calls(str(odd))["dp"].tolist() == [5]
    empty = tmp_path / "empty.vcf"
    empty.write_text(head)
    _same_calls(_vcf.read_calls(str(empty), native=True), _vcf.read_calls(str(empty), native=False))


def test_native_vcf_reader_in_many_blocks_on_several_threads(golden_dir, tmp_path, monkeypatch):
    """the native reader cuts the file into blocks of whole lines parsed by a team of threads: with 8-KiB blocks the sample file
    becomes hundreds of blocks, and the records must come back in file order, equal to the Python reader's, for any team size"""
    import gzip
    import shutil
    src = os.path.join(golden_dir, "701_501.filter.vcf.gz")
    plain = str(tmp_path / "s.vcf")
    with gzip.open(src, "rb") as fi, open(plain, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    want = _vcf.read_calls(plain, (0,), native=False)
    for threads, block_kb in (("1", "8"), ("4", "8"), ("16", "4"), ("3", "4096")):
        monkeypatch.setenv("SNPM_VCF_THREADS", threads)
        monkeypatch.setenv("SNPM_VCF_BLOCK_KB", block_kb)
        for path in (plain, src):
            got = _vcf.read_calls(path, (0,))
            assert got["called"] is not None and np.array_equal(got["called"], (want["gt"][:, 0] != "./.") & (want["gt"][:, 0] != ".|."))
            _same_calls(got, want)


def test_vcf_weights_carry_dictionary_codes(golden_dir, tmp_path):
    """a parsed VCF keeps (codes, table) with table[codes] == wei bit for bit (what Genotyper's one-call path uploads instead of the
    fp64 triples); the cache round-trips them, an edit of the weights drops them"""
    import gzip
    import shutil
    src = os.path.join(golden_dir, "701_501.filter.vcf.gz")
    vcf = str(tmp_path / "s.vcf.gz")
    shutil.copy(src, vcf)
    p = parsers.ParseInputs(vcf)
    p.wait_for_cache()
    pair = p.weight_codes()
    assert pair is not None
    codes, table = pair
    assert codes.dtype == np.uint16 and codes.shape == p.wei.shape == (7545, 3)
    assert np.array_equal(table[codes].view(np.uint64), p.wei.view(np.uint64))
    assert table[0] == 1.0 and 0.0 in table and not p.wei.flags.writeable
    with pytest.raises(ValueError):
        p.wei[0, 0] = 0.5                                  # in-place edits cannot leave the codes stale
    q = parsers.ParseInputs(vcf)                           # from the .npz cache
    assert q.weight_codes() is not None and np.array_equal(q.weight_codes()[0], codes) and np.array_equal(q.wei, p.wei)
    q.wei = q.wei.copy()                                   # a new weight array: no codes any more
    assert q.weight_codes() is None
    hard = parsers.ParseInputs("")
    hard.load_snp_info(["1"], [5], ["0/0"], np.array([[1.0, 0.0, 0.0]]), 3)
    assert hard.weight_codes() is None                     # nothing parsed, nothing coded
    # fractional or huge PLs are not coded
    assert parsers._weight_codes(np.array([[0.0, 10.5, 200.0]]), np.array([False]), np.exp(np.array([[0.0, 10.5, 200.0]]) / -10)) is None
    assert parsers._weight_codes(np.array([[0.0, 70000.0, 200.0]]), np.array([False]), np.exp(np.array([[0.0, 70000.0, 200.0]]) / -10)) is None
    both = parsers._weight_codes(np.array([[0.0, -1.0, 30.0], [-1.0, -1.0, -1.0]]), np.array([False, True]),
                                 np.array([[1.0, np.exp(0.1), np.exp(-3.0)], [0.0, 1.0, 0.0]]))
    if both is not None:                                   # (exp(0.1) as a one-element array may round differently: then no codes)
        assert np.array_equal(both[1][both[0]].view(np.uint64), np.array([[1.0, np.exp(0.1), np.exp(-3.0)], [0.0, 1.0, 0.0]]).view(np.uint64))


def _write_bgzf(path, data, member=65280, eof=True):
    """``data`` as a BGZF file (what bgzip / bcftools write): gzip members of at most ``member`` text bytes, each with its
    compressed size in a 'BC' extra subfield; the empty end-of-file member last"""
    import struct
    import zlib
    with open(path, "wb") as fh:
        pieces = [data[i:i + member] for i in range(0, len(data), member)] + ([b""] if eof else [])
        for piece in pieces:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            raw = co.compress(piece) + co.flush()
            fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25))
            fh.write(raw + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))


def test_native_vcf_reader_inflates_bgzf_members_side_by_side(golden_dir, tmp_path, monkeypatch):
    """a bgzipped VCF (independent gzip members) is inflated by a team of threads, batch by batch: same records as the plain
    file for any member size, batch size and team size; gzip's own reader agrees on the file; damaged members are refused"""
    import gzip
    from snpmatch_amd import _lib
    src = os.path.join(golden_dir, "701_501.filter.vcf.gz")
    data = gzip.open(src, "rb").read()
    plain = str(tmp_path / "s.vcf")
    open(plain, "wb").write(data)
    want = _vcf.read_calls(plain, (0,), native=False)
    bg = str(tmp_path / "s.bgzf.vcf.gz")
    for member, batch_kb, threads, block_kb in ((65280, "16384", "8", "4096"), (4000, "1", "4", "8"), (517, "3", "16", "4"),
                                                (65280, "70", "1", "64"), (30000, "64", "3", "8")):
        _write_bgzf(bg, data, member)
        assert gzip.open(bg, "rb").read() == data                   # a valid multi-member gzip file
        monkeypatch.setenv("SNPM_VCF_BGZF_BATCH_KB", batch_kb)
        monkeypatch.setenv("SNPM_VCF_THREADS", threads)
        monkeypatch.setenv("SNPM_VCF_BLOCK_KB", block_kb)
        _same_calls(_vcf.read_calls(bg, (0,), native=True), want)
        monkeypatch.setenv("SNPM_VCF_BGZF", "0")                    # the one-stream path on the same file
        _same_calls(_vcf.read_calls(bg, (0,), native=True), want)
        monkeypatch.delenv("SNPM_VCF_BGZF")
    for k in ("SNPM_VCF_BGZF_BATCH_KB", "SNPM_VCF_THREADS", "SNPM_VCF_BLOCK_KB"):
        monkeypatch.delenv(k)
    # a batch is bounded by its INFLATED size as well (192 MiB; here 40 / 300 KiB): the members that do not fit wait for the next
    # refill -- also when a single member is larger than the compressed batch, and when the cap is smaller than one member
    for member, batch_kb, text_kb in ((65280, "16384", "300"), (65280, "16", "40"), (2000, "64", "40"), (30000, "1", "1")):
        _write_bgzf(bg, data, member)
        monkeypatch.setenv("SNPM_VCF_BGZF_BATCH_KB", batch_kb)
        monkeypatch.setenv("SNPM_VCF_BGZF_TEXT_KB", text_kb)
        _same_calls(_vcf.read_calls(bg, (0,), native=True), want)
    monkeypatch.delenv("SNPM_VCF_BGZF_BATCH_KB")
    monkeypatch.delenv("SNPM_VCF_BGZF_TEXT_KB")
    _write_bgzf(bg, data, 65280, eof=False)                          # no end-of-file member: still every record
    _same_calls(_vcf.read_calls(bg, (0,), native=True), want)
    # damage: a flipped byte inside a member's data (CRC / inflate error), a truncated file, trailing bytes that are no member
    _write_bgzf(bg, data, 20000)
    raw = bytearray(open(bg, "rb").read())
    bad = bytes(raw[:5000]) + bytes([raw[5000] ^ 0x55]) + bytes(raw[5001:])
    open(bg, "wb").write(bad)
    assert _lib.vcf_parse(bg, 0) is None
    open(bg, "wb").write(bytes(raw[:len(raw) // 2]))
    assert _lib.vcf_parse(bg, 0) is None
    open(bg, "wb").write(bytes(raw) + b"garbage-that-is-no-member")
    assert _lib.vcf_parse(bg, 0) is None


def test_parse_cache_is_read_through_one_memory_map(tmp_path):
    """the .npz parse cache (np.savez: stored members) is opened as views of one memory map instead of through zipfile; the arrays
    equal np.load's, and anything else (compressed members, object arrays) is left to np.load"""
    rng = np.random.default_rng(3)
    n = 5000
    pl = rng.integers(0, 200, size=(n, 3)).astype(float)
    wei = np.exp(pl / -10)
    p = parsers.ParseInputs("")
    p.load_snp_info(np.array(["Chr%d" % (1 + i * 5 // n) for i in range(n)]), np.arange(n) * 7 + 1,
                    rng.choice(np.array(["0/0", "1/1", "0/1", "./."]), size=n), wei, rng.integers(1, 40, n))
    p._adopt_codes(parsers._weight_codes(pl, np.zeros(n, bool), wei))
    path = str(tmp_path / "s.vcf.snpmatch")
    p.save_snp_info(path)
    fast, slow = parsers._StoredNpz.open(path + ".npz"), np.load(path + ".npz")
    assert fast is not None and sorted(fast.files) == sorted(slow.files)
    for k in slow.files:
        assert fast[k].dtype == slow[k].dtype and fast[k].shape == slow[k].shape and np.array_equal(fast[k], slow[k]), k
    q = parsers.ParseInputs("")
    q._load_npz(path + ".npz")
    for k in ("chrs", "pos", "gt", "wei", "dp"):
        assert np.array_equal(getattr(q, k), getattr(p, k)), k
    assert q.weight_codes() is not None and np.array_equal(q.weight_codes()[0], p.weight_codes()[0])
    # BED-style cache: dp is the string "NA" (a 0-d member), no codes stored
    np.savez(str(tmp_path / "bed.npz"), chr=p.chrs, pos=p.pos, gt=p.gt, wei=np.eye(3)[rng.integers(0, 3, n)], dp="NA")
    b = parsers.ParseInputs("")
    b._load_npz(str(tmp_path / "bed.npz"))
    assert str(b.dp) == "NA" and b.weight_codes() is not None and b.weight_codes()[1].tolist() == [0.0, 1.0]
    np.savez_compressed(str(tmp_path / "c.npz"), a=np.arange(5))
    assert parsers._StoredNpz.open(str(tmp_path / "c.npz")) is None
    np.savez(str(tmp_path / "o.npz"), a=np.array([{"x": 1}], dtype=object))
    assert parsers._StoredNpz.open(str(tmp_path / "o.npz")) is None
    assert parsers._StoredNpz.open(str(tmp_path / "missing.npz")) is None


def test_gt_codes_of_equals_parse_gt_of_the_gathered_strings():
    """ParseInputs.gt_codes_of(rows) == parseGT(gt[rows]) -- including parseGT's habit of taking the separator from the FIRST entry
    it is given (a mixed '/' and '|' column parses differently depending on which row leads); cached per separator and per array"""
    gt = np.array(["0/0", "0/1", "1|1", "./.", "1/0", "0|1", ".|.", "1/1"] * 7)
    p = parsers.ParseInputs("")
    p.load_snp_info(["1"] * len(gt), np.arange(len(gt)), gt, np.zeros((len(gt), 3)), 3)
    rng = np.random.default_rng(2)
    for _ in range(30):
        rows = rng.choice(len(gt), size=int(rng.integers(1, 40)), replace=False)
        assert np.array_equal(p.gt_codes_of(rows), parsers.parseGT(p.gt[rows])), rows
    assert len(p.gt_codes_of(np.zeros(0, dtype=int))) == 0
    cached = p._gt_code_cache["/"][1]
    p.gt_codes_of(np.array([0, 1]))
    assert p._gt_code_cache["/"][1] is cached                     # the column is parsed once per separator
    p.gt = np.array(["1/1"] * len(gt))                            # a new column: parsed again
    assert p.gt_codes_of(np.array([3, 4])).tolist() == [1, 1]
    q = parsers.ParseInputs("")
    q.load_snp_info(["1", "1"], [1, 2], np.array(["0", "1"]), np.zeros((2, 3)), 3)      # numeric codes: no separator, the caller parses
    assert q.gt_codes_of(np.array([0, 1])) is None


def test_read_vcf_semantics_on_hand_built_records(tmp_path, monkeypatch):
    """ParseInputs.read_vcf as the reference states it (core/parsers.py:141-157; scikit-allel is in neither interpreter of this
    image, so the rules are pinned on records written by hand): records whose GT is './.' or '.|.' are dropped (:144); PL goes
    through exp(PL / -10) (:148-149); a record WITHOUT any PL (allel's fill value -1 in all three places, :147) takes the hard 0/1
    weights of its genotype (:150, get_wei_from_GT :132-139: ref -> column 0, het -> column 1, alt -> column 2); depth is INFO/DP
    (:155, import_vcf_file :203-206), fill value -1"""
    head = ("##fileformat=VCFv4.2\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"depth\">\n"
            "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"genotype\">\n##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"PL\">\n"
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")
    body = [
        "Chr1\t100\t.\tA\tT\t50\tPASS\tDP=7\tGT:PL\t0/0:0,30,255",
        "Chr1\t200\t.\tA\tT\t50\tPASS\tDP=9\tGT:PL\t./.:.",                # no call: dropped
        "Chr1\t300\t.\tA\tT\t50\tPASS\tDP=3\tGT\t1/1",                      # no PL: hard weights of the genotype
        "Chr1\t400\t.\tA\tT\t50\tPASS\t.\tGT:PL\t0/1:20,0,40",             # no INFO/DP: -1
        "Chr2\t500\t.\tA\tT\t50\tPASS\tDP=5\tGT:PL\t.|.:0,0,0",             # phased no call: dropped
        "Chr2\t600\t.\tA\tT\t.\t.\tDP=2\tGT:DP:PL\t1/1:2:99,12,0",          # PL as the third FORMAT item
        "Chr2\t700\t.\tA\tT\t.\t.\tDP=4\tGT\t0/0",                          # no PL again
    ]
    vcf = tmp_path / "hand.vcf"
    vcf.write_text(head + "\n".join(body) + "\n")
    for native in (True, False):
        if not native:
            monkeypatch.setattr(_vcf, "_read_calls_native", lambda path, sample: None)      # the generic Python reader
        if os.path.exists(str(vcf) + ".snpmatch.npz"):
            os.remove(str(vcf) + ".snpmatch.npz")
        s = parsers.ParseInputs(inFile=str(vcf), logDebug=False)
        s.wait_for_cache()
        assert s.chrs.tolist() == ["Chr1", "Chr1", "Chr1", "Chr2", "Chr2"] and s.pos.tolist() == [100, 300, 400, 600, 700]
        assert s.gt.tolist() == ["0/0", "1/1", "0/1", "1/1", "0/0"]
        want = np.array([np.exp(np.array([0.0, 30.0, 255.0]) / (-10)), [0.0, 0.0, 1.0], np.exp(np.array([20.0, 0.0, 40.0]) / (-10)),
                         np.exp(np.array([99.0, 12.0, 0.0]) / (-10)), [1.0, 0.0, 0.0]])
        assert np.array_equal(s.wei.view(np.uint64), want.view(np.uint64)), native
        assert np.asarray(s.dp).tolist() == [7, 3, -1, 2, 4]
