"""Kaldi object / table / model file formats on either side of the hot path (SURVEY §8f
rows 2 and 4): what `nnet-latgen-faster` reads (features ark/scp, `final.mdl` =
TransitionModel + AmNnet, `HCLG.fst`) and writes (lattices, alignments, words).

Host-side, numpy only — the formats are byte layouts, not compute.  Restated from

  base/io-funcs{.h,-inl.h,.cc}        tokens, basic types, integer vectors, "\\0B" header
  matrix/kaldi-matrix.cc:1155-1345    Matrix Read/Write ("FM " / "DM ", text " [ ... ]")
  matrix/kaldi-vector.cc:1040-1170    Vector Read/Write ("FV " / "DV ")
  matrix/compressed-matrix.cc:27-37,244-250,363-373,404-530   "CM " / "CM2 "
  util/kaldi-table-inl.h, util/kaldi-holder-inl.h              ark / scp tables
  nnet2/nnet-nnet.cc:160-189, am-nnet.cc:31-42, nnet-component.cc (Read of every
      component the forward path supports)
  gmm/am-diag-gmm.cc:147-176, gmm/diag-gmm.cc:705-756   AmDiagGmm / DiagGmm
  hmm/hmm-topology.cc:39-196, hmm/transition-model.cc:72-98,274-320
  lat/kaldi-lattice.cc:394-430        lattice text / binary (OpenFst VectorFst) I/O

Pinned (tests/test_kaldi_io.py, fixtures written by the REFERENCE's own Write functions
compiled in oracle/_ref): matrices, vectors, compressed matrices, integer vectors, ark/scp
tables, Nnet / AmNnet and AmDiagGmm in binary and text mode, HmmTopology.  PARITY UNPINNED (the
reference's code for them needs OpenFst 1.3.4, absent here): TransitionModel's wrapper
tokens, the OpenFst binary FST layout (HCLG read, lattice write) and FstPrinter's text
lattice layout — restated from the reference's call sites and OpenFst's published format.
"""
import io
import os
import struct

import numpy as np

# ---------------------------------------------------------------- stream primitives


class Stream:
    """Byte stream with an exact peek (std::istream as io-funcs uses it): own look-ahead
    buffer, since io.BufferedReader.peek(n) may return fewer than n bytes."""

    def __init__(self, f):
        self.f = f
        self.buf = b""
        self.pos = 0          # bytes consumed since construction
        try:
            self.base = f.tell()
        except (OSError, AttributeError, io.UnsupportedOperation):
            self.base = 0

    def _fill(self, n):
        while len(self.buf) < n:
            chunk = self.f.read(max(n - len(self.buf), 1 << 16))
            if not chunk:
                break
            self.buf += chunk

    def peek(self, n=1):
        self._fill(n)
        return self.buf[:n]

    def get(self, n=1):
        self._fill(n)
        if len(self.buf) < n:
            raise EOFError("unexpected end of Kaldi stream")
        out, self.buf = self.buf[:n], self.buf[n:]
        self.pos += n
        return out

    def tell(self):
        return self.base + self.pos

    def skip_ws(self):
        while True:
            b = self.peek()
            if b and b in b" \t\n\r":
                self.get()
            else:
                return

    def eof(self):
        return len(self.peek()) == 0


def _as_stream(f):
    return f if isinstance(f, Stream) else Stream(f)


def init_kaldi_input(s):
    """InitKaldiInputStream (io-funcs-inl.h:284-301): "\\0B" => binary."""
    if s.peek(2) == b"\0B":
        s.get(2)
        return True
    return False


def read_token(s, binary=True):
    """ReadToken (io-funcs.cc:154-167): whitespace-delimited word, one trailing space consumed."""
    s.skip_ws()
    out = bytearray()
    while True:
        b = s.peek()
        if not b:
            break
        if b in b" \t\n\r":
            s.get()
            break
        out += s.get()
    if not out:
        raise EOFError("ReadToken: end of stream")
    return out.decode("latin-1")


def expect_token(s, binary, token):
    got = read_token(s, binary)
    if got != token:
        raise ValueError("Expected token \"%s\", got instead \"%s\"." % (token, got))


def expect_one_or_two_tokens(s, binary, token1, token2):
    """ExpectOneOrTwoTokens (nnet2/nnet-component.cc): token1 is optional."""
    got = read_token(s, binary)
    if got == token1:
        expect_token(s, binary, token2)
    elif got != token2:
        raise ValueError("Expecting token %s or %s but got %s" % (token1, token2, got))


def _text_word(s):
    s.skip_ws()
    out = bytearray()
    while True:
        b = s.peek()
        if not b or b in b" \t\n\r":
            break
        out += s.get()
    return out.decode("latin-1")


def read_int32(s, binary=True):
    """ReadBasicType<int32> (io-funcs-inl.h:72-102)."""
    if binary:
        n = s.get()[0]
        if n != 4:
            raise ValueError("ReadBasicType: expected size 4, saw %d" % n)
        return struct.unpack("<i", s.get(4))[0]
    return int(_text_word(s))


def read_float(s, binary=True):
    """ReadBasicType<float/double> (io-funcs.cc:82-151): the size byte says which."""
    if binary:
        n = s.get()[0]
        if n == 4:
            return struct.unpack("<f", s.get(4))[0]
        if n == 8:
            return struct.unpack("<d", s.get(8))[0]
        raise ValueError("ReadBasicType: expected float, saw size %d" % n)
    w = _text_word(s).lower()
    return float({"inf": "inf", "infinity": "inf", "-inf": "-inf", "-infinity": "-inf", "nan": "nan"}.get(w, w))


def read_bool(s, binary=True):
    if not binary:
        s.skip_ws()
    c = s.get()
    if c not in (b"T", b"F"):
        raise ValueError("Read failure in ReadBasicType<bool>, saw %r" % c)
    return c == b"T"


def read_int32_vector(s, binary=True):
    """ReadIntegerVector (io-funcs-inl.h:226-281)."""
    if binary:
        n = s.get()[0]
        if n != 4:
            raise ValueError("ReadIntegerVector: expected size 4, saw %d" % n)
        size = struct.unpack("<i", s.get(4))[0]
        return np.frombuffer(s.get(4 * size), dtype="<i4").copy() if size else np.zeros(0, np.int32)
    s.skip_ws()
    if s.get() != b"[":
        raise ValueError("ReadIntegerVector: expected to see [")
    out = []
    while True:
        s.skip_ws()
        if s.peek() == b"]":
            s.get()
            break
        out.append(int(_text_word_until_bracket(s)))
    return np.asarray(out, np.int32)


def _text_word_until_bracket(s):
    out = bytearray()
    while True:
        b = s.peek()
        if not b or b in b" \t\n\r]":
            break
        out += s.get()
    return out.decode("latin-1")


def write_token(f, binary, token):
    f.write(token.encode("latin-1") + b" ")


def write_int32(f, binary, v):
    f.write(b"\x04" + struct.pack("<i", int(v)) if binary else b"%d " % int(v))


def write_float(f, binary, v):
    f.write(b"\x04" + struct.pack("<f", float(v)) if binary else (_fmt(v) + " ").encode())


def write_int32_vector(f, binary, v):
    v = np.ascontiguousarray(v, "<i4")
    if binary:
        f.write(b"\x04" + struct.pack("<i", len(v)) + v.tobytes())
    else:
        f.write(b"[ " + b"".join(b"%d " % int(x) for x in v) + b"]\n")


# ---------------------------------------------------------------- matrices and vectors


def _read_text_rows(s):
    """Text matrix body after the opening '[' up to ']' (kaldi-matrix.cc:1216-1345)."""
    rows, cur = [], []
    word = bytearray()

    def flush_word():
        if word:
            w = word.decode("latin-1").lower()
            cur.append(float({"infinity": "inf", "-infinity": "-inf"}.get(w, w)))
            word.clear()
    while True:
        b = s.get()
        if b == b"]":
            flush_word()
            if cur:
                rows.append(cur)
            break
        if b in b"\n;":
            flush_word()
            if cur:
                rows.append(cur)
                cur = []
        elif b in b" \t\r":
            flush_word()
        else:
            word += b
    # the trailing newline after ']' is left for the next skip_ws
    return rows


def read_matrix(s, binary=True):
    """Matrix<Real>::Read; also accepts a CompressedMatrix (as CompressedMatrix::Read does
    the converse).  Returns float32 (or float64 for "DM")."""
    s = _as_stream(s)
    if binary:
        if s.peek() == b"C":
            return read_compressed_matrix(s, binary)
        tok = read_token(s, binary)
        if tok not in ("FM", "DM"):
            raise ValueError("Expected token FM or DM, got " + tok)
        rows, cols = read_int32(s), read_int32(s)
        dt = "<f4" if tok == "FM" else "<f8"
        n = rows * cols * (4 if tok == "FM" else 8)
        return np.frombuffer(s.get(n), dtype=dt).reshape(rows, cols).copy() if n else np.zeros((rows, cols), dt)
    s.skip_ws()
    if s.get() != b"[":
        raise ValueError("Expected \"[\" at the start of a text matrix")
    rows = _read_text_rows(s)
    if not rows:
        return np.zeros((0, 0), np.float32)
    if len({len(r) for r in rows}) != 1:
        raise ValueError("Matrix has inconsistent #cols")
    return np.asarray(rows, np.float32)


def read_vector(s, binary=True):
    s = _as_stream(s)
    if binary:
        tok = read_token(s, binary)
        if tok not in ("FV", "DV"):
            raise ValueError("Expected token FV or DV, got " + tok)
        dim = read_int32(s)
        dt, w = ("<f4", 4) if tok == "FV" else ("<f8", 8)
        return np.frombuffer(s.get(dim * w), dtype=dt).copy() if dim else np.zeros(0, dt)
    s.skip_ws()
    if s.get() != b"[":
        raise ValueError("Expected \"[\" at the start of a text vector")
    rows = _read_text_rows(s)
    return np.asarray(rows[0] if rows else [], np.float32)


def _fmt(x):
    # operator<< of a float at the precision InitKaldiOutputStream sets (io-funcs-inl.h:274-281: 7 digits)
    x = float(x)
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    return "%.7g" % x


def write_matrix(f, M, binary=True):
    """Matrix<float>::Write (kaldi-matrix.cc:1155-1193)."""
    M = np.asarray(M)
    dt, tok = ("<f8", b"DM ") if M.dtype == np.float64 else ("<f4", b"FM ")
    M = np.ascontiguousarray(M, dt)
    if binary:
        f.write(tok + b"\x04" + struct.pack("<i", M.shape[0]) + b"\x04" + struct.pack("<i", M.shape[1]) + M.tobytes())
    elif M.shape[1] == 0 or M.shape[0] == 0:
        f.write(b" [ ]\n")
    else:
        f.write(b" [")
        for r in M:
            f.write(b"\n  " + b"".join((_fmt(x) + " ").encode() for x in r))
        f.write(b"]\n")


def write_vector(f, v, binary=True):
    v = np.asarray(v)
    dt, tok = ("<f8", b"DV ") if v.dtype == np.float64 else ("<f4", b"FV ")
    v = np.ascontiguousarray(v, dt)
    if binary:
        f.write(tok + b"\x04" + struct.pack("<i", len(v)) + v.tobytes())
    else:
        f.write(b" [ " + b"".join((_fmt(x) + " ").encode() for x in v) + b"]\n")


def read_compressed_matrix(s, binary=True):
    """CompressedMatrix::Read + CopyToMat (compressed-matrix.cc:436-530)."""
    s = _as_stream(s)
    tok = read_token(s, binary)
    if tok not in ("CM", "CM2"):
        raise ValueError("Unexpected token %s, expecting CM or CM2." % tok)
    min_value, rng, rows, cols = struct.unpack("<ffii", s.get(16))
    if cols == 0:
        return np.zeros((0, 0), np.float32)
    f32 = np.float32

    def u16_to_float(v):  # Uint16ToFloat :244-250, float arithmetic in the reference's order
        return f32(min_value) + f32(rng) * f32(1.52590218966964e-05) * v.astype(np.float32)
    if tok == "CM2":
        data = np.frombuffer(s.get(2 * rows * cols), dtype="<u2").reshape(rows, cols)
        return u16_to_float(data).astype(np.float32)
    hdr = np.frombuffer(s.get(8 * cols), dtype="<u2").reshape(cols, 4)
    p = u16_to_float(hdr)                                  # [cols, 4] p0 p25 p75 p100
    b = np.frombuffer(s.get(rows * cols), dtype=np.uint8).reshape(cols, rows)   # column-major bytes
    v = b.astype(np.float32)
    p0, p25, p75, p100 = (p[:, k:k + 1] for k in range(4))
    # CharToFloat :363-373: float * uchar * double constant, rounded to float on return
    lo = (p0.astype(np.float64) + ((p25 - p0) * v).astype(np.float64) * (1 / 64.0))
    mid = (p25.astype(np.float64) + ((p75 - p25) * (v - f32(64))).astype(np.float64) * (1 / 128.0))
    hi = (p75.astype(np.float64) + ((p100 - p75) * (v - f32(192))).astype(np.float64) * (1 / 63.0))
    out = np.where(b <= 64, lo, np.where(b <= 192, mid, hi)).astype(np.float32)
    return np.ascontiguousarray(out.T)


# ---------------------------------------------------------------- tables (ark / scp)

def _read_object(s, binary, kind):
    if kind == "matrix":
        return read_matrix(s, binary)
    if kind == "vector":
        return read_vector(s, binary)
    if kind == "int32_vector":
        # BasicVectorHolder<int32>::Read (kaldi-holder-inl.h:229-280): NOT WriteIntegerVector's layout —
        # binary: the size, then every element as a basic type; text: one line "1 2 3 \n"
        if binary:
            return np.asarray([read_int32(s) for _ in range(read_int32(s))], np.int32)
        line = bytearray()
        while not s.eof():
            b = s.get()
            if b == b"\n":
                break
            line += b
        return np.asarray([int(w) for w in line.split()], np.int32)
    if kind == "compact_lattice":
        return read_compact_lattice(s, binary)
    if kind == "lattice":
        return read_lattice(s, binary)
    if kind == "any_lattice":
        return read_any_lattice(s, binary)
    if kind == "posterior":
        return read_posterior(s, binary)
    raise ValueError("unknown table object kind " + kind)


def _write_object(f, binary, kind, obj):
    if kind == "matrix":
        write_matrix(f, obj, binary)
    elif kind == "vector":
        write_vector(f, obj, binary)
    elif kind == "int32_vector":        # BasicVectorHolder<int32>::Write (kaldi-holder-inl.h:197-224)
        v = np.asarray(obj, np.int32)
        if binary:
            write_int32(f, True, len(v))
            for x in v:
                write_int32(f, True, x)
        else:
            f.write(b"".join(b"%d " % int(x) for x in v) + b"\n")
    elif kind == "lattice":
        write_lattice(f, obj, binary)
    elif kind == "compact_lattice":
        write_compact_lattice(f, obj, binary)
    elif kind == "posterior":
        write_posterior(f, obj, binary)
    elif kind == "base_float":
        write_float(f, binary, obj)
        if not binary:
            f.write(b"\n")
    else:
        raise ValueError("unknown table object kind " + kind)


def read_fst_holder(s):
    """fst::VectorFstHolder::Read (fstext/fstext-utils.h:416-520), the object of a table of decoding graphs
    (`SequentialTableReader<fst::VectorFstHolder>`, nnet-latgen-faster.cc:141): NO "\\0B" header - the binary form is the
    OpenFst file itself (it starts with the magic number, never with a space), the text form starts with a newline,
    holds one arc "src dst ilabel olabel [weight]" or one final state "state [weight]" per line and ends with an empty
    line.  Returns the CSR graph dict of read_fst."""
    s = _as_stream(s)
    c = s.peek()
    if not c:
        raise EOFError("End of stream detected reading Fst")
    if c not in b" \t\r\n":
        return read_fst(s)
    while s.peek() in (b" ", b"\t", b"\r"):
        s.get()
    if s.peek() != b"\n":
        raise ValueError("Reading FST: unexpected sequence of spaces")
    s.get()
    arcs, finals, n_states, start = [], {}, 0, None
    while True:
        line = bytearray()
        while not s.eof():
            b = s.get()
            if b == b"\n":
                break
            line += b
        col = line.decode().split()
        if not col:
            break          # the terminating empty line (or the end of the stream)
        if len(col) > 5 or len(col) == 3:   # (3 columns: "not ok ...; it's not an acceptor", fstext-utils.h:479-481 - the read fails)
            raise ValueError("Bad line in FST: " + line.decode())
        src = int(col[0])
        if start is None:
            start = src    # "the first state mentioned is the start state" (:455-458)
        n_states = max(n_states, src + 1)
        if len(col) <= 2:   # final state [weight]
            finals[src] = float(col[1]) if len(col) == 2 else 0.0
        else:
            dst = int(col[1])
            n_states = max(n_states, dst + 1)
            il = int(col[2])
            ol = int(col[3])
            w = float(col[4]) if len(col) == 5 else 0.0
            arcs.append((src, dst, il, ol, w))
    order = sorted(range(len(arcs)), key=lambda i: arcs[i][0])   # stable: arcs of a state keep their order
    off = np.zeros(n_states + 1, np.int64)
    for a in arcs:
        off[a[0] + 1] += 1
    off = np.cumsum(off)
    fin = np.full(n_states, np.inf, np.float32)
    for st, w in finals.items():
        fin[st] = w
    return dict(num_states=n_states, start=0 if start is None else start, arc_offsets=off,
                ilabel=np.asarray([arcs[i][2] for i in order], np.int32), olabel=np.asarray([arcs[i][3] for i in order], np.int32),
                weight=np.asarray([arcs[i][4] for i in order], np.float32), nextstate=np.asarray([arcs[i][1] for i in order], np.int32),
                final=fin)


def write_fst_holder(f, g, binary):
    """fst::VectorFstHolder::Write (fstext/fstext-utils.h:378-408): see read_fst_holder."""
    if binary:
        write_fst(f, g)
        return
    off = np.asarray(g["arc_offsets"], np.int64)
    out = [b"\n"]
    states = [int(g["start"])] + [st for st in range(int(g["num_states"])) if st != int(g["start"])]   # the start state first
    for st in states:
        for a in range(int(off[st]), int(off[st + 1])):
            out.append(b"%d\t%d\t%d\t%d\t%r\n" % (st, int(g["nextstate"][a]), int(g["ilabel"][a]), int(g["olabel"][a]),
                                                   float(np.float32(g["weight"][a]))))
        if np.isfinite(g["final"][st]):
            out.append(b"%d\t%r\n" % (st, float(np.float32(g["final"][st]))))
    out.append(b"\n")
    f.write(b"".join(out))


def read_ark(path_or_file, kind="matrix"):
    """SequentialTableReader over an archive (kaldi-table-inl.h:340-520): yields (key, object)."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
    s = Stream(f)
    try:
        while True:
            s.skip_ws()
            if s.eof():
                return
            key = read_token(s, False)
            binary = init_kaldi_input(s)
            yield key, _read_object(s, binary, kind)
    finally:
        if isinstance(path_or_file, (str, os.PathLike)):
            f.close()


def read_scp(path, kind="matrix"):
    """Script-file reader (kaldi-table-inl.h:63-340): lines "key rxfilename[:offset]"."""
    with open(path, "r") as sf:
        for line in sf:
            line = line.strip()
            if not line:
                continue
            key, rx = line.split(None, 1)
            offset = 0
            head, sep, tail = rx.rpartition(":")
            if sep and tail.isdigit():
                rx, offset = head, int(tail)
            with open(rx, "rb") as f:
                f.seek(offset)
                s = Stream(f)
                binary = init_kaldi_input(s)
                yield key, _read_object(s, binary, kind)


class TableWriter:
    """TableWriter for "ark:", "ark,t:", "ark,scp:" wspecifiers (kaldi-table-inl.h:760-1100)."""

    def __init__(self, ark_path, scp_path=None, kind="matrix", binary=True):
        self.f = open(ark_path, "wb")
        self.ark_path = ark_path
        self.scp = open(scp_path, "w") if scp_path else None
        self.kind, self.binary = kind, binary

    def write(self, key, obj):
        if not key or any(c.isspace() for c in key):
            raise ValueError("invalid table key %r" % key)
        self.f.write(key.encode() + b" ")
        if self.scp:
            self.scp.write("%s %s:%d\n" % (key, self.ark_path, self.f.tell()))
        if self.binary:
            self.f.write(b"\0B")
        _write_object(self.f, self.binary, self.kind, obj)

    def close(self):
        self.f.close()
        if self.scp:
            self.scp.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---------------------------------------------------------------- nnet2 models

def _skip_scalars_until(s, binary, end_token):
    """After the parameters of an updatable component: <Token> scalar pairs up to the end token."""
    out = {}
    while True:
        tok = read_token(s, binary)
        if tok == end_token:
            return out
        if binary:
            b = s.peek()
            if b in (b"T", b"F"):
                out[tok] = read_bool(s, binary)
            elif b == b"\x04" or b == b"\x08":
                n = s.get()[0]
                raw = s.get(n)
                out[tok] = raw
            else:
                raise ValueError("unexpected byte %r after %s" % (b, tok))
        else:
            out[tok] = _text_word(s)


def _read_nonlinear(s, binary, typ):
    """NonlinearComponent::Read (nnet-component.cc:369-392)."""
    end = "</%s>" % typ
    expect_one_or_two_tokens(s, binary, "<%s>" % typ, "<Dim>")
    dim = read_int32(s, binary)
    tok = read_token(s, binary)
    if tok == "<ValueSum>":
        read_vector(s, binary)
        expect_token(s, binary, "<DerivSum>")
        read_vector(s, binary)
        expect_token(s, binary, "<Count>")
        read_float(s, binary)
        expect_token(s, binary, end)
    elif tok == "<Counts>":
        read_vector(s, binary)
        expect_token(s, binary, end)
    elif tok != end:
        raise ValueError("expected %s, got %s" % (end, tok))
    return dim


def read_component(s, binary, already_read_type=None):
    """Component::ReadNew (nnet-component.cc:140-165) for the component types the forward
    path implements; returns the dict api.Nnet takes."""
    typ = already_read_type or read_token(s, binary)
    typ = typ.strip("<>")
    if typ in ("AffineComponent", "AffineComponentPreconditioned", "AffineComponentPreconditionedOnline"):
        expect_token(s, binary, "<LearningRate>")
        read_float(s, binary)
        expect_token(s, binary, "<LinearParams>")
        W = read_matrix(s, binary)
        expect_token(s, binary, "<BiasParams>")
        b = read_vector(s, binary)
        if typ == "AffineComponent":        # :1256-1286 (with the <AvgInput> back-compatibility branch)
            tok = read_token(s, binary)
            if tok == "<AvgInput>":
                read_vector(s, binary)
                expect_token(s, binary, "<AvgInputCount>")
                read_float(s, binary)
                tok = read_token(s, binary)
            if tok == "<IsGradient>":
                read_bool(s, binary)
                expect_token(s, binary, "</%s>" % typ)
            elif tok != "</%s>" % typ:
                raise ValueError("unexpected token " + tok)
        else:                               # :1417-1445, :1665-1703: scalar configuration values
            _skip_scalars_until(s, binary, "</%s>" % typ)
        return dict(type="affine", input_dim=W.shape[1], output_dim=W.shape[0], linear=W.astype(np.float32),
                    bias=b.astype(np.float32), kaldi_type=typ)
    if typ == "FixedAffineComponent":       # :3373-3379
        expect_token(s, binary, "<LinearParams>")
        W = read_matrix(s, binary)
        expect_token(s, binary, "<BiasParams>")
        b = read_vector(s, binary)
        expect_token(s, binary, "</FixedAffineComponent>")
        return dict(type="fixed_affine", input_dim=W.shape[1], output_dim=W.shape[0], linear=W.astype(np.float32),
                    bias=b.astype(np.float32))
    if typ == "SpliceComponent":            # :2814-2837
        expect_token(s, binary, "<InputDim>")
        input_dim = read_int32(s, binary)
        tok = read_token(s, binary)
        if tok == "<LeftContext>":
            left = read_int32(s, binary)
            expect_token(s, binary, "<RightContext>")
            right = read_int32(s, binary)
            context = np.arange(-left, right + 1, dtype=np.int32)
        elif tok == "<Context>":
            context = read_int32_vector(s, binary)
        else:
            raise ValueError("Unknown token %s, the model might be corrupted" % tok)
        expect_token(s, binary, "<ConstComponentDim>")
        const_dim = read_int32(s, binary)
        expect_token(s, binary, "</SpliceComponent>")
        return dict(type="splice", input_dim=input_dim, context=context, const_dim=const_dim,
                    output_dim=(input_dim - const_dim) * len(context) + const_dim)
    if typ == "PnormComponent":             # :542-550
        expect_token(s, binary, "<InputDim>")
        i = read_int32(s, binary)
        expect_token(s, binary, "<OutputDim>")
        o = read_int32(s, binary)
        expect_token(s, binary, "<P>")
        p = read_float(s, binary)
        expect_token(s, binary, "</PnormComponent>")
        return dict(type="pnorm", input_dim=i, output_dim=o, p=p)
    if typ in ("NormalizeComponent", "SoftmaxComponent"):
        dim = _read_nonlinear(s, binary, typ)
        return dict(type="normalize" if typ == "NormalizeComponent" else "softmax", input_dim=dim, output_dim=dim)
    if typ == "SumGroupComponent":          # :2455-2467
        expect_token(s, binary, "<Sizes>")
        sizes = read_int32_vector(s, binary)
        tok = read_token(s, binary)
        if tok not in ("<SumGroupComponent>", "</SumGroupComponent>"):
            raise ValueError("Expected </SumGroupComponent>, got " + tok)
        return dict(type="sum_group", input_dim=int(sizes.sum()), output_dim=len(sizes), sizes=sizes)
    if typ == "FixedScaleComponent":        # :3446-3450
        expect_token(s, binary, "<Scales>")
        v = read_vector(s, binary)
        expect_token(s, binary, "</FixedScaleComponent>")
        return dict(type="fixed_scale", input_dim=len(v), output_dim=len(v), bias=v.astype(np.float32))
    if typ == "FixedBiasComponent":         # :3515-3519
        expect_token(s, binary, "<Bias>")
        v = read_vector(s, binary)
        expect_token(s, binary, "</FixedBiasComponent>")
        return dict(type="fixed_bias", input_dim=len(v), output_dim=len(v), bias=v.astype(np.float32))
    raise NotImplementedError("component type %s is not on the implemented forward path" % typ)


def read_nnet(s, binary):
    """Nnet::Read (nnet-nnet.cc:175-189)."""
    expect_token(s, binary, "<Nnet>")
    expect_token(s, binary, "<NumComponents>")
    n = read_int32(s, binary)
    expect_token(s, binary, "<Components>")
    comps = [read_component(s, binary) for _ in range(n)]
    expect_token(s, binary, "</Components>")
    expect_token(s, binary, "</Nnet>")
    for a, b in zip(comps[:-1], comps[1:]):     # Nnet::Check
        if a["output_dim"] != b["input_dim"]:
            raise ValueError("component dimension mismatch: %d vs %d" % (a["output_dim"], b["input_dim"]))
    return comps


def read_am_nnet(s, binary):
    """AmNnet::Read (am-nnet.cc:39-42): the Nnet, then the priors."""
    comps = read_nnet(s, binary)
    priors = read_vector(s, binary).astype(np.float32)
    return comps, priors


def read_topology(s, binary):
    """HmmTopology::Read (hmm-topology.cc:39-139).  Returns dict(phones, phone2idx, entries)
    with entries[i][j] = (pdf_class, [(dst_state, prob), ...])."""
    expect_token(s, binary, "<Topology>")
    phones, phone2idx, entries = [], [], []
    if binary:
        phones = [int(x) for x in read_int32_vector(s, True)]
        phone2idx = [int(x) for x in read_int32_vector(s, True)]
        for _ in range(read_int32(s)):
            entry = []
            for _ in range(read_int32(s)):
                pdf_class = read_int32(s)
                trans = []
                for _ in range(read_int32(s)):
                    dst = read_int32(s)
                    trans.append((dst, read_float(s)))
                entry.append((pdf_class, trans))
            entries.append(entry)
        expect_token(s, binary, "</Topology>")
    else:
        while True:
            tok = read_token(s, False)
            if tok == "</Topology>":
                break
            if tok != "<TopologyEntry>":
                raise ValueError("Reading HmmTopology object, expected </Topology> or <TopologyEntry>, got " + tok)
            expect_token(s, False, "<ForPhones>")
            these = []
            while True:
                w = read_token(s, False)
                if w == "</ForPhones>":
                    break
                these.append(int(w))
            entry = []
            tok = read_token(s, False)
            while tok != "</TopologyEntry>":
                if tok != "<State>":
                    raise ValueError("Expected </TopologyEntry> or <State>, got instead " + tok)
                state = read_int32(s, False)
                if state != len(entry):
                    raise ValueError("States are expected to be in order from zero")
                tok = read_token(s, False)
                pdf_class = -1
                if tok == "<PdfClass>":
                    pdf_class = read_int32(s, False)
                    tok = read_token(s, False)
                trans = []
                while tok == "<Transition>":
                    dst = read_int32(s, False)
                    trans.append((dst, float(np.float32(read_float(s, False)))))   # BaseFloat
                    tok = read_token(s, False)
                if tok != "</State>":
                    raise ValueError("Reading HmmTopology,  unexpected token " + tok)
                entry.append((pdf_class, trans))
                tok = read_token(s, False)
            idx = len(entries)
            entries.append(entry)
            for ph in these:
                if ph <= 0:
                    raise ValueError("phone ids are positive")
                if len(phone2idx) <= ph:
                    phone2idx += [-1] * (ph + 1 - len(phone2idx))
                if phone2idx[ph] != -1:
                    raise ValueError("Phone appears in multiple topology entries.")
                phone2idx[ph] = idx
                phones.append(ph)
        phones.sort()
    return dict(phones=phones, phone2idx=phone2idx, entries=entries)


def read_transition_model(s, binary):
    """TransitionModel::Read + ComputeDerived (transition-model.cc:274-294, 72-98).
    Returns dict(topo, triples [n,3], log_probs, tid2pdf) — tid2pdf[tid] = TransitionIdToPdf(tid)
    (:index 0 unused, -1), which is all the decoder needs (decodable-am-nnet.h:60-69)."""
    expect_token(s, binary, "<TransitionModel>")
    topo = read_topology(s, binary)
    expect_token(s, binary, "<Triples>")
    n = read_int32(s, binary)
    triples = np.asarray([[read_int32(s, binary) for _ in range(3)] for _ in range(n)], np.int32).reshape(n, 3)
    expect_token(s, binary, "</Triples>")
    expect_token(s, binary, "<LogProbs>")
    log_probs = read_vector(s, binary)
    expect_token(s, binary, "</LogProbs>")
    expect_token(s, binary, "</TransitionModel>")
    # ComputeDerived :72-98: transition-ids are numbered triple by triple, transition by transition of the HMM-state's topology entry
    tid2pdf, tid2phone, tid2hmm, tid_self = [-1], [0], [-1], [False]
    for phone, hmm_state, pdf in triples:
        entry = topo["entries"][topo["phone2idx"][phone]]
        trans = entry[hmm_state][1]
        tid2pdf += [int(pdf)] * len(trans)
        tid2phone += [int(phone)] * len(trans)                       # TransitionIdToPhone :235-239
        tid2hmm += [int(hmm_state)] * len(trans)                     # TransitionIdToHmmState :247-251
        tid_self += [int(dst) == int(hmm_state) for dst, _ in trans]  # IsSelfLoop :217-225
    if len(tid2pdf) != len(log_probs):
        raise ValueError("TransitionModel: %d transition-ids but %d log-probs" % (len(tid2pdf) - 1, len(log_probs)))
    return dict(topo=topo, triples=triples, log_probs=log_probs.astype(np.float32), tid2pdf=np.asarray(tid2pdf, np.int32),
                tid2phone=np.asarray(tid2phone, np.int32), tid2hmm_state=np.asarray(tid2hmm, np.int32),
                tid_is_self_loop=np.asarray(tid_self, bool))


def read_diag_gmm(s, binary):
    """DiagGmm::Read (gmm/diag-gmm.cc:728-756).  The stored gconsts are read and dropped, as the
    reference recomputes them ("safer option than trusting the read gconsts")."""
    tok = read_token(s, binary)
    if tok not in ("<DiagGMMBegin>", "<DiagGMM>"):
        raise ValueError("Expected <DiagGMM>, got " + tok)
    tok = read_token(s, binary)
    if tok == "<GCONSTS>":
        read_vector(s, binary)
        expect_token(s, binary, "<WEIGHTS>")
    elif tok != "<WEIGHTS>":
        raise ValueError("DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " + tok)
    weights = read_vector(s, binary).astype(np.float32)
    expect_token(s, binary, "<MEANS_INVVARS>")
    means_invvars = read_matrix(s, binary).astype(np.float32)
    expect_token(s, binary, "<INV_VARS>")
    inv_vars = read_matrix(s, binary).astype(np.float32)
    tok = read_token(s, binary)
    if tok not in ("<DiagGMMEnd>", "</DiagGMM>"):
        raise ValueError("Expected </DiagGMM>, got " + tok)
    return weights, means_invvars, inv_vars


def read_am_diag_gmm(s, binary):
    """AmDiagGmm::Read (gmm/am-diag-gmm.cc:147-161) -> the concatenated arrays api.AmDiagGmm takes:
    dict(weights [M], means_invvars [M, D], inv_vars [M, D], pdf_offsets [num_pdfs + 1], dim)."""
    expect_token(s, binary, "<DIMENSION>")
    dim = read_int32(s, binary)
    expect_token(s, binary, "<NUMPDFS>")
    num_pdfs = read_int32(s, binary)
    if num_pdfs <= 0:
        raise ValueError("AmDiagGmm: num_pdfs > 0")
    w, mi, iv, off = [], [], [], [0]
    for _ in range(num_pdfs):
        a, b, c = read_diag_gmm(s, binary)
        if b.shape[1] != dim or c.shape != b.shape or len(a) != len(b):
            raise ValueError("AmDiagGmm: inconsistent DiagGmm dimensions")
        w.append(a); mi.append(b); iv.append(c)
        off.append(off[-1] + len(a))
    return dict(weights=np.concatenate(w), means_invvars=np.concatenate(mi, 0), inv_vars=np.concatenate(iv, 0),
                pdf_offsets=np.asarray(off, np.int32), dim=dim)


def read_gmm_model(path):
    """`final.mdl` as gmm-latgen-faster reads it (gmmbin/gmm-latgen-faster.cc:76-83):
    TransitionModel, then AmDiagGmm."""
    with open(path, "rb") as f:
        s = Stream(f)
        binary = init_kaldi_input(s)
        tm = read_transition_model(s, binary)
        am = read_am_diag_gmm(s, binary)
    return tm, am


def write_topology(f, topo, binary=True):
    """HmmTopology::Write, binary branch (hmm-topology.cc:176-193)."""
    if not binary:
        raise NotImplementedError("text-mode topology writing")
    write_token(f, True, "<Topology>")
    write_int32_vector(f, True, topo["phones"])
    write_int32_vector(f, True, topo["phone2idx"])
    write_int32(f, True, len(topo["entries"]))
    for entry in topo["entries"]:
        write_int32(f, True, len(entry))
        for pdf_class, trans in entry:
            write_int32(f, True, pdf_class)
            write_int32(f, True, len(trans))
            for dst, prob in trans:
                write_int32(f, True, dst)
                write_float(f, True, prob)
    write_token(f, True, "</Topology>")


def write_transition_model(f, topo, triples, log_probs, binary=True):
    """TransitionModel::Write (transition-model.cc:296-320), binary."""
    write_token(f, binary, "<TransitionModel>")
    write_topology(f, topo, binary)
    write_token(f, binary, "<Triples>")
    write_int32(f, binary, len(triples))
    for t in triples:
        for x in t:
            write_int32(f, binary, x)
    write_token(f, binary, "</Triples>")
    write_token(f, binary, "<LogProbs>")
    write_vector(f, np.asarray(log_probs, np.float32), binary)
    write_token(f, binary, "</LogProbs>")
    write_token(f, binary, "</TransitionModel>")


def read_nnet2_model(path):
    """`final.mdl` as nnet-latgen-faster reads it (nnet2bin/nnet-latgen-faster.cc:86-93):
    TransitionModel, then AmNnet.  Returns (trans_model, components, priors)."""
    with open(path, "rb") as f:
        s = Stream(f)
        binary = init_kaldi_input(s)
        tm = read_transition_model(s, binary)
        comps, priors = read_am_nnet(s, binary)
    return tm, comps, priors


# ---------------------------------------------------------------- OpenFst binary FSTs

_FST_MAGIC = 2125659606
_SYMTAB_MAGIC = 2125658996


def _fst_string(s):
    n = struct.unpack("<i", s.get(4))[0]
    return s.get(n).decode("latin-1")


def _skip_symbol_table(s):
    if struct.unpack("<i", s.get(4))[0] != _SYMTAB_MAGIC:
        raise ValueError("bad symbol table magic")
    _fst_string(s)
    _, size = struct.unpack("<qq", s.get(16))
    for _ in range(size):
        _fst_string(s)
        s.get(8)


def _read_fst_header(s):
    if struct.unpack("<i", s.get(4))[0] != _FST_MAGIC:
        raise ValueError("not an OpenFst binary FST (bad magic)")
    fsttype, arctype = _fst_string(s), _fst_string(s)
    version, flags = struct.unpack("<ii", s.get(8))
    props, start, nstates, narcs = struct.unpack("<Qqqq", s.get(32))
    if flags & 1:
        _skip_symbol_table(s)
    if flags & 2:
        _skip_symbol_table(s)
    return dict(fsttype=fsttype, arctype=arctype, version=version, flags=flags, start=start, num_states=nstates,
                num_arcs=narcs)


def read_fst(path_or_stream):
    """fst::Fst<StdArc>::Read as `ReadFstKaldi` does for HCLG.fst (fstext/fstext-utils-inl.h,
    nnet-latgen-faster.cc:107): VectorFst or ConstFst, arc type "standard" (tropical, float)
    or "lattice4" (LatticeWeight: two floats).  Returns the CSR graph dict api.Fst takes
    (+ "weight2" for lattice arcs).  PARITY UNPINNED: OpenFst is absent here."""
    own = isinstance(path_or_stream, (str, os.PathLike))
    f = open(path_or_stream, "rb") if own else None
    s = Stream(f) if own else _as_stream(path_or_stream)
    try:
        h = _read_fst_header(s)
        nw = {"standard": 1, "lattice4": 2}.get(h["arctype"])
        if nw is None:
            raise NotImplementedError("arc type " + h["arctype"])
        wdt = [("w", "<f4")] if nw == 1 else [("w", "<f4"), ("w2", "<f4")]
        arc_dt = np.dtype([("il", "<i4"), ("ol", "<i4")] + wdt + [("ns", "<i4")])
        ns = h["num_states"]
        if h["fsttype"] == "vector":
            finals, counts, chunks = [], [], []
            for _ in range(ns):
                finals.append(struct.unpack("<" + "f" * nw, s.get(4 * nw)))
                n = struct.unpack("<q", s.get(8))[0]
                counts.append(n)
                chunks.append(s.get(arc_dt.itemsize * n))
            arcs = np.frombuffer(b"".join(chunks), dtype=arc_dt)
            finals = np.asarray(finals, np.float32).reshape(ns, nw)
            off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        elif h["fsttype"] == "const":
            aligned = h["version"] == 1 or (h["flags"] & 4)

            def align():
                if aligned:
                    pad = (-s.tell()) % 16
                    if pad:
                        s.get(pad)
            st_dt = np.dtype([("final", "<f4", (nw,)), ("pos", "<u4"), ("narcs", "<u4"), ("nie", "<u4"), ("noe", "<u4")])
            align()
            st = np.frombuffer(s.get(st_dt.itemsize * ns), dtype=st_dt)
            align()
            arcs = np.frombuffer(s.get(arc_dt.itemsize * h["num_arcs"]), dtype=arc_dt)
            finals = st["final"].reshape(ns, nw).astype(np.float32)
            off = np.concatenate([st["pos"].astype(np.int64), [h["num_arcs"]]])
        else:
            raise NotImplementedError("fst type " + h["fsttype"])
        g = dict(num_states=int(ns), start=int(h["start"]), arc_offsets=off, ilabel=arcs["il"].astype(np.int32),
                 olabel=arcs["ol"].astype(np.int32), weight=arcs["w"].astype(np.float32),
                 nextstate=arcs["ns"].astype(np.int32), final=np.ascontiguousarray(finals[:, 0]), tid2pdf=None)
        if nw == 2:
            g["weight2"] = arcs["w2"].astype(np.float32)
            g["final2"] = np.ascontiguousarray(finals[:, 1])
        return g
    finally:
        if own:
            f.close()


def _lattice_csr(lat):
    """The raw-lattice dict of api.LatticeFasterDecoder.get_raw_lattice as (offsets, arcs sorted by source)."""
    n = len(lat["state_frame"])
    src = np.asarray(lat["arc_src"], np.int64)
    order = np.argsort(src, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))]).astype(np.int64)
    return n, off, order


def write_lattice(f, lat, binary=True):
    """WriteLattice (lat/kaldi-lattice.cc:394-430): binary = VectorFst<LatticeArc>::Write
    (arc type "lattice4", LatticeWeight = (graph, acoustic)); text = a newline, FstPrinter's
    tab-separated lines (weight "graph,acoustic", omitted when it is One), a blank line.
    State 0 is the start state (GetRawLattice :128-132).  PARITY UNPINNED (OpenFst absent)."""
    n, off, order = _lattice_csr(lat)
    il, ol = np.asarray(lat["arc_il"], np.int32), np.asarray(lat["arc_ol"], np.int32)
    g, a = np.asarray(lat["arc_g"], np.float32), np.asarray(lat["arc_a"], np.float32)
    dst = np.asarray(lat["arc_dst"], np.int32)
    fin = np.asarray(lat["state_final"], np.float32)
    inf = np.float32(np.inf)
    if binary:
        def fst_str(x):
            return struct.pack("<i", len(x)) + x
        props = 0x1 | 0x2   # kExpanded | kMutable; everything else "unknown"
        f.write(struct.pack("<i", _FST_MAGIC) + fst_str(b"vector") + fst_str(b"lattice4") + struct.pack("<ii", 2, 0) +
                struct.pack("<Qqqq", props, 0 if n else -1, n, len(dst)))
        arc_dt = np.dtype([("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("w2", "<f4"), ("ns", "<i4")])
        arcs = np.zeros(len(dst), arc_dt)
        arcs["il"], arcs["ol"], arcs["w"], arcs["w2"], arcs["ns"] = il[order], ol[order], g[order], a[order], dst[order]
        for st in range(n):
            f.write(struct.pack("<ff", fin[st], np.float32(0.0) if fin[st] != inf else inf))
            f.write(struct.pack("<q", off[st + 1] - off[st]))
            f.write(arcs[off[st]:off[st + 1]].tobytes())
        return
    f.write(b"\n")

    def wstr(v1, v2):
        return "%s,%s" % (_fst_float(v1), _fst_float(v2))
    for st in range(n):
        for k in order[off[st]:off[st + 1]]:
            line = "%d\t%d\t%d\t%d" % (st, dst[k], il[k], ol[k])
            if not (g[k] == 0.0 and a[k] == 0.0):
                line += "\t" + wstr(g[k], a[k])
            f.write(line.encode() + b"\n")
        if fin[st] != inf:
            f.write(("%d" % st if fin[st] == 0.0 else "%d\t%s" % (st, wstr(fin[st], 0.0))).encode() + b"\n")
    f.write(b"\n")


def _fst_float(x):
    x = float(x)
    if x == float("inf"):
        return "Infinity"
    if x == float("-inf"):
        return "-Infinity"
    return "%.7g" % x


def read_lattice(s, binary=True):
    """ReadLattice (kaldi-lattice.cc:300-392) into the raw-lattice dict layout (without the
    decoder-side state_frame / state_hclg annotations, which a lattice file does not hold)."""
    s = _as_stream(s)
    if binary:
        g = read_fst(s)
        if "weight2" not in g:
            raise ValueError("not a lattice (arc type is not lattice4)")
        n = g["num_states"]
        src = np.repeat(np.arange(n, dtype=np.int32), np.diff(g["arc_offsets"]))
        fin = np.where((g["final"] == np.inf) | (g["final2"] == np.inf), np.float32(np.inf), g["final"] + g["final2"])
        return dict(num_states=n, start=g["start"], arc_src=src, arc_dst=g["nextstate"], arc_il=g["ilabel"],
                    arc_ol=g["olabel"], arc_g=g["weight"], arc_a=g["weight2"], state_final=fin.astype(np.float32),
                    state_final_graph=g["final"], state_final_acoustic=g["final2"])
    arcs, finals, nstates, start = [], {}, 0, None

    def weight(tok):
        a, _, b = tok.partition(",")
        conv = lambda t: float({"Infinity": "inf", "-Infinity": "-inf"}.get(t, t))
        return conv(a), conv(b)
    # the FST is terminated by an empty line
    first = True
    while True:
        line = bytearray()
        while True:
            b = s.get() if not s.eof() else b"\n"
            if b == b"\n":
                break
            line += b
        txt = line.decode().strip()
        if not txt:
            if first:
                first = False
                continue
            break
        first = False
        col = txt.split()
        if len(col) >= 4:
            w = weight(col[4]) if len(col) > 4 else (0.0, 0.0)
            arcs.append((int(col[0]), int(col[1]), int(col[2]), int(col[3]), w[0], w[1]))
            nstates = max(nstates, int(col[0]) + 1, int(col[1]) + 1)
        else:
            finals[int(col[0])] = weight(col[1]) if len(col) > 1 else (0.0, 0.0)
            nstates = max(nstates, int(col[0]) + 1)
        if start is None:
            start = int(col[0])
    A = np.asarray(arcs, np.float64).reshape(-1, 6)
    fg = np.full(nstates, np.inf, np.float32)
    fa = np.full(nstates, np.inf, np.float32)
    for st, (x, y) in finals.items():
        fg[st], fa[st] = x, y
    return dict(num_states=nstates, start=0 if start is None else start, arc_src=A[:, 0].astype(np.int32),
                arc_dst=A[:, 1].astype(np.int32), arc_il=A[:, 2].astype(np.int32), arc_ol=A[:, 3].astype(np.int32),
                arc_g=A[:, 4].astype(np.float32), arc_a=A[:, 5].astype(np.float32),
                state_final=np.where(fg == np.inf, np.float32(np.inf), fg + fa).astype(np.float32),
                state_final_graph=fg, state_final_acoustic=fa)


def compact_lattice_to_lattice(c):
    """ConvertLattice(CompactLattice -> Lattice, invert = true) fstext/lattice-utils-inl.h:114-186: an arc's transition-ids
    become a chain of arcs (input side), its word and weight sit on the first of them; a final weight's string a chain to a
    new final state.  Returns the raw-lattice dict layout (state 0 = the CompactLattice's state 0)."""
    n = int(c["n_states"])
    src, dst, il, ol, g, a = [], [], [], [], [], []
    fin_g = np.full(n, np.inf, np.float32)
    fin_a = np.full(n, np.inf, np.float32)
    fin_g, fin_a = fin_g.tolist(), fin_a.tolist()

    def add_state():
        fin_g.append(float("inf"))
        fin_a.append(float("inf"))
        return len(fin_g) - 1
    for s in range(n):
        if np.isfinite(c["final_g"][s]) and np.isfinite(c["final_a"][s]):
            string = [int(x) for x in c["final_string"][s]]
            cur = s
            for k, t in enumerate(string):
                nx = add_state()
                src.append(cur); dst.append(nx); il.append(t); ol.append(0)
                g.append(float(c["final_g"][s]) if k == 0 else 0.0); a.append(float(c["final_a"][s]) if k == 0 else 0.0)
                cur = nx
            fin_g[cur], fin_a[cur] = (0.0, 0.0) if string else (float(c["final_g"][s]), float(c["final_a"][s]))
    for j in range(len(c["arc_src"])):
        string = [int(x) for x in c["arc_string"][j]]
        cur, L = int(c["arc_src"][j]), len(string)
        for k in range(L - 1):
            nx = add_state()
            src.append(cur); dst.append(nx); il.append(string[k]); ol.append(int(c["arc_label"][j]) if k == 0 else 0)
            g.append(float(c["arc_g"][j]) if k == 0 else 0.0); a.append(float(c["arc_a"][j]) if k == 0 else 0.0)
            cur = nx
        src.append(cur); dst.append(int(c["arc_dst"][j])); il.append(string[-1] if L else 0)
        ol.append(int(c["arc_label"][j]) if L <= 1 else 0)
        g.append(float(c["arc_g"][j]) if L <= 1 else 0.0); a.append(float(c["arc_a"][j]) if L <= 1 else 0.0)
    fg, fa = np.asarray(fin_g, np.float32), np.asarray(fin_a, np.float32)
    return dict(num_states=len(fin_g), start=0, arc_src=np.asarray(src, np.int32), arc_dst=np.asarray(dst, np.int32),
                arc_il=np.asarray(il, np.int32), arc_ol=np.asarray(ol, np.int32), arc_g=np.asarray(g, np.float32),
                arc_a=np.asarray(a, np.float32), state_final=np.where(np.isinf(fg) | np.isinf(fa), np.float32(np.inf), fg + fa).astype(np.float32),
                state_final_graph=fg, state_final_acoustic=fa)


def read_any_lattice(s, binary=True):
    """LatticeHolder::Read (lat/kaldi-lattice.cc:394-430): the table may hold Lattices or CompactLattices (what
    `lattice-to-post "ark:gunzip -c lat.1.gz|"` reads is the decoder's CompactLattice output); either comes back as a
    state-level lattice."""
    s = _as_stream(s)
    if binary:
        head = s.peek(64)
        n1 = struct.unpack("<i", head[4:8])[0]
        n2 = struct.unpack("<i", head[8 + n1:12 + n1])[0]
        arctype = head[12 + n1:12 + n1 + n2].decode()
        if arctype.startswith("compactlattice"):
            return compact_lattice_to_lattice(read_compact_lattice(s, True))
        return read_lattice(s, True)
    # text: a CompactLattice is an acceptor - "src dst word g,a,1_2_3" - so its 4th column is the weight (it has commas);
    # the 4th column of a Lattice line is the output label
    look = s.peek(4096).split(b"\n")
    for line in look:
        col = line.split()
        if len(col) >= 4:
            return compact_lattice_to_lattice(read_compact_lattice(s, False)) if b"," in col[3] else read_lattice(s, False)
    return read_lattice(s, False)


def write_posterior(f, post, binary=True):
    """PosteriorHolder::Write hmm/posterior.cc:31-66; post = per frame a list of (int32, float)."""
    if binary:
        f.write(b"\4" + struct.pack("<i", len(post)))
        for frame in post:
            f.write(b"\4" + struct.pack("<i", len(frame)))
            for i, w in frame:
                f.write(b"\4" + struct.pack("<i", int(i)) + b"\4" + struct.pack("<f", float(w)))
    else:
        out = []
        for frame in post:
            out.append("[ " + "".join("%d %s " % (int(i), _fmt(np.float32(w))) for i, w in frame) + "] ")
        f.write(("".join(out) + "\n").encode())


def read_posterior(s, binary=True):
    """PosteriorHolder::Read hmm/posterior.cc:68-140."""
    s = _as_stream(s)
    if binary:
        out = []
        for _ in range(read_int32(s)):
            n = read_int32(s)
            out.append([(read_int32(s), read_float(s)) for _ in range(n)])
        return out
    line = bytearray()
    while not s.eof():
        b = s.get()
        if b == b"\n":
            break
        line += b
    out, cur = [], None
    tok = line.decode().split()
    k = 0
    while k < len(tok):
        if tok[k] == "[":
            cur = []
            k += 1
        elif tok[k] == "]":
            out.append(cur)
            cur = None
            k += 1
        else:
            cur.append((int(tok[k]), float(tok[k + 1])))
            k += 2
    return out


def write_compact_lattice(f, clat, binary=True):
    """WriteCompactLattice (lat/kaldi-lattice.cc:366-392) of api.determinize_lattice_pruned's dict:
    binary = VectorFst<CompactLatticeArc>::Write, arc type "compactlattice44", weight =
    LatticeWeight (two floats) + int32 string length + int32 transition-ids
    (fstext/lattice-weight.h:503-511); text = FstPrinter lines "src dst word word g,a,t1_t2_..."
    (acceptor: one label column; weight omitted when it is One), fstext/lattice-weight.h:676-686.
    PARITY UNPINNED (OpenFst absent): round trip + hand-assembled known answers."""
    n = int(clat["n_states"])
    src = np.asarray(clat["arc_src"], np.int64)
    order = np.argsort(src, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))]).astype(np.int64)
    inf = np.float32(np.inf)
    if binary:
        def fst_str(x):
            return struct.pack("<i", len(x)) + x
        props = 0x1 | 0x2
        f.write(struct.pack("<i", _FST_MAGIC) + fst_str(b"vector") + fst_str(b"compactlattice44") + struct.pack("<ii", 2, 0) +
                struct.pack("<Qqqq", props, 0 if n else -1, n, len(src)))

        def weight(g, a, string):
            string = np.asarray(string, np.int32)
            return struct.pack("<ffi", g, a, len(string)) + string.astype("<i4").tobytes()
        for st in range(n):
            if clat["final_g"][st] != inf:
                f.write(weight(clat["final_g"][st], clat["final_a"][st], clat["final_string"][st]))
            else:
                f.write(weight(inf, inf, []))
            f.write(struct.pack("<q", off[st + 1] - off[st]))
            for k in order[off[st]:off[st + 1]]:
                f.write(struct.pack("<ii", clat["arc_label"][k], clat["arc_label"][k]) +
                        weight(clat["arc_g"][k], clat["arc_a"][k], clat["arc_string"][k]) + struct.pack("<i", clat["arc_dst"][k]))
        return
    f.write(b"\n")

    def wstr(g, a, string):
        return "%s,%s,%s" % (_fst_float(g), _fst_float(a), "_".join(str(int(x)) for x in string))
    for st in range(n):
        for k in order[off[st]:off[st + 1]]:
            line = "%d\t%d\t%d" % (st, clat["arc_dst"][k], clat["arc_label"][k])
            if not (clat["arc_g"][k] == 0.0 and clat["arc_a"][k] == 0.0 and len(clat["arc_string"][k]) == 0):
                line += "\t" + wstr(clat["arc_g"][k], clat["arc_a"][k], clat["arc_string"][k])
            f.write(line.encode() + b"\n")
        if clat["final_g"][st] != inf:
            one = clat["final_g"][st] == 0.0 and clat["final_a"][st] == 0.0 and len(clat["final_string"][st]) == 0
            f.write(("%d" % st if one else "%d\t%s" % (st, wstr(clat["final_g"][st], clat["final_a"][st], clat["final_string"][st]))).encode() + b"\n")
    f.write(b"\n")


def read_compact_lattice(s, binary=True):
    """ReadCompactLattice (lat/kaldi-lattice.cc:330-364) into api.determinize_lattice_pruned's layout."""
    s = _as_stream(s)
    if binary:
        h = {}
        magic, = struct.unpack("<i", s.get(4))
        if magic != _FST_MAGIC:
            raise ValueError("not an OpenFst file")
        def rstr():
            n, = struct.unpack("<i", s.get(4))
            return s.get(n)
        fsttype, arctype = rstr(), rstr()
        if fsttype != b"vector" or arctype != b"compactlattice44":
            raise ValueError("not a compact lattice: %r %r" % (fsttype, arctype))
        version, flags = struct.unpack("<ii", s.get(8))
        props, start, n, narcs = struct.unpack("<Qqqq", s.get(32))

        def weight():
            g, a, k = struct.unpack("<ffi", s.get(12))
            return g, a, np.frombuffer(s.get(4 * k), "<i4").astype(np.int32)
        fg, fa, fs = np.empty(n, np.float32), np.empty(n, np.float32), []
        asrc, adst, alab, ag, aa, astr = [], [], [], [], [], []
        for st in range(n):
            g, a, string = weight()
            fg[st], fa[st] = g, a
            fs.append(string)
            k, = struct.unpack("<q", s.get(8))
            for _ in range(k):
                il, ol = struct.unpack("<ii", s.get(8))
                g, a, string = weight()
                d, = struct.unpack("<i", s.get(4))
                asrc.append(st); adst.append(d); alab.append(il); ag.append(g); aa.append(a); astr.append(string)
        return dict(n_states=int(n), arc_src=np.array(asrc, np.int32), arc_dst=np.array(adst, np.int32),
                    arc_label=np.array(alab, np.int32), arc_g=np.array(ag, np.float32), arc_a=np.array(aa, np.float32),
                    arc_string=astr, final_g=fg, final_a=fa, final_string=fs, complete=True)
    arcs, finals, nstates = [], {}, 0

    def weight(tok):
        parts = tok.split(",")
        conv = lambda t: float({"Infinity": "inf", "-Infinity": "-inf"}.get(t, t))
        string = np.array([int(x) for x in parts[2].split("_")] if len(parts) > 2 and parts[2] else [], np.int32)
        return conv(parts[0]), conv(parts[1]), string
    first = True
    while True:
        line = bytearray()
        while True:
            b = s.get() if not s.eof() else b"\n"
            if b == b"\n":
                break
            line += b
        txt = line.decode().strip()
        if not txt:
            if first:
                first = False
                continue
            break
        first = False
        col = txt.split()
        if len(col) >= 3:
            g, a, string = weight(col[3]) if len(col) > 3 else (0.0, 0.0, np.zeros(0, np.int32))
            arcs.append((int(col[0]), int(col[1]), int(col[2]), g, a, string))
            nstates = max(nstates, int(col[0]) + 1, int(col[1]) + 1)
        else:
            finals[int(col[0])] = weight(col[1]) if len(col) > 1 else (0.0, 0.0, np.zeros(0, np.int32))
            nstates = max(nstates, int(col[0]) + 1)
    fg = np.full(nstates, np.inf, np.float32)
    fa = np.full(nstates, np.inf, np.float32)
    fs = [np.zeros(0, np.int32) for _ in range(nstates)]
    for st, (g, a, string) in finals.items():
        fg[st], fa[st], fs[st] = g, a, string
    return dict(n_states=nstates, arc_src=np.array([x[0] for x in arcs], np.int32), arc_dst=np.array([x[1] for x in arcs], np.int32),
                arc_label=np.array([x[2] for x in arcs], np.int32), arc_g=np.array([x[3] for x in arcs], np.float32),
                arc_a=np.array([x[4] for x in arcs], np.float32), arc_string=[x[5] for x in arcs], final_g=fg, final_a=fa,
                final_string=fs, complete=True)


def write_fst(f, g):
    """VectorFst<StdArc>::Write of a CSR graph dict (the inverse of read_fst; used to build
    HCLG test files).  PARITY UNPINNED."""
    def fst_str(x):
        return struct.pack("<i", len(x)) + x
    n = int(g["num_states"])
    off = np.asarray(g["arc_offsets"], np.int64)
    f.write(struct.pack("<i", _FST_MAGIC) + fst_str(b"vector") + fst_str(b"standard") + struct.pack("<ii", 2, 0) +
            struct.pack("<Qqqq", 3, int(g["start"]), n, int(off[-1])))
    arc_dt = np.dtype([("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("ns", "<i4")])
    arcs = np.zeros(int(off[-1]), arc_dt)
    arcs["il"], arcs["ol"], arcs["w"], arcs["ns"] = g["ilabel"], g["olabel"], g["weight"], g["nextstate"]
    fin = np.asarray(g["final"], np.float32)
    for st in range(n):
        f.write(struct.pack("<f", fin[st]) + struct.pack("<q", off[st + 1] - off[st]) + arcs[off[st]:off[st + 1]].tobytes())


# ---------------------------------------------------------------- online2 front-end files
def read_sp_matrix(s, binary=True):
    """SpMatrix<Real> = PackedMatrix<Real>::Read (matrix/packed-matrix.cc:132-234): binary "FP"/"DP" +
    int32 rows + the lower triangle by rows; text "[ r0 \\n r1 r1 ... ]".  Returns the FULL symmetric
    matrix (float64 for "DP")."""
    s = _as_stream(s)
    if binary:
        tok = read_token(s, binary)
        if tok not in ("FP", "DP"):
            raise ValueError("Expected token FP or DP, got " + tok)
        n = read_int32(s)
        dt, w = ("<f4", 4) if tok == "FP" else ("<f8", 8)
        m = n * (n + 1) // 2
        packed = np.frombuffer(s.get(m * w), dtype=dt)
    else:
        s.skip_ws()
        if s.get() != b"[":
            raise ValueError("Expected \"[\" at the start of a text packed matrix")
        rows = _read_text_rows(s)
        n = len(rows)
        if any(len(r) != i + 1 for i, r in enumerate(rows)):
            raise ValueError("PackedMatrix: row i must hold i + 1 elements")
        packed = np.asarray([x for r in rows for x in r], np.float64)
    out = np.zeros((n, n), packed.dtype)
    r, c = np.tril_indices(n)
    out[r, c] = packed
    out[c, r] = packed
    return out


def write_sp_matrix(f, M, binary=True):
    """PackedMatrix<Real>::Write (packed-matrix.cc:236-271) of the lower triangle of M."""
    M = np.asarray(M)
    dt, tok = ("<f8", b"DP ") if M.dtype == np.float64 else ("<f4", b"FP ")
    n = M.shape[0]
    r, c = np.tril_indices(n)
    packed = np.ascontiguousarray(M[r, c], dt)
    if binary:
        f.write(tok + b"\x04" + struct.pack("<i", n) + packed.tobytes())
    elif n == 0:
        f.write(b"[ ]\n")
    else:
        f.write(b"[\n")
        i = 0
        for j in range(n):
            f.write(b"".join((_fmt(x) + " ").encode() for x in packed[i:i + j + 1]))
            i += j + 1
            f.write(b"]\n" if j == n - 1 else b"\n")


def read_double(s, binary=True):
    """ReadBasicType<double> (io-funcs-inl.h: size byte 8 + the value; text: a word)."""
    if binary:
        if s.get() != b"\x08":
            raise ValueError("ReadBasicType: expected a double")
        return struct.unpack("<d", s.get(8))[0]
    return float(_text_word(s))


def write_double(f, binary, v):
    f.write(b"\x08" + struct.pack("<d", float(v)) if binary else (repr(float(v)) + " ").encode())


def write_diag_gmm(f, weights, means_invvars, inv_vars, binary=True):
    """DiagGmm::Write (gmm/diag-gmm.cc:705-720); the gconsts are recomputed as ComputeGconsts does
    (in float: the reader drops them anyway)."""
    w = np.asarray(weights, np.float32)
    mi = np.asarray(means_invvars, np.float32)
    iv = np.asarray(inv_vars, np.float32)
    D = mi.shape[1]
    g = np.log(w.astype(np.float64)) - 0.5 * (D * np.log(2 * np.pi) + np.sum(-np.log(iv.astype(np.float64)) +
                                                                           mi.astype(np.float64) ** 2 / iv.astype(np.float64), 1))
    write_token(f, binary, "<DiagGMM>")
    if not binary:
        f.write(b"\n")
    write_token(f, binary, "<GCONSTS>")
    write_vector(f, g.astype(np.float32), binary)
    write_token(f, binary, "<WEIGHTS>")
    write_vector(f, w, binary)
    write_token(f, binary, "<MEANS_INVVARS>")
    write_matrix(f, mi, binary)
    write_token(f, binary, "<INV_VARS>")
    write_matrix(f, iv, binary)
    write_token(f, binary, "</DiagGMM>")
    if not binary:
        f.write(b"\n")


def write_am_diag_gmm(f, am, binary=True):
    """AmDiagGmm::Write (gmm/am-diag-gmm.cc:163-175): <DIMENSION> dim <NUMPDFS> n, then every DiagGmm.
    `am`: the dict read_am_diag_gmm returns."""
    off = np.asarray(am["pdf_offsets"])
    write_token(f, binary, "<DIMENSION>")
    write_int32(f, binary, int(am["dim"]))
    write_token(f, binary, "<NUMPDFS>")
    write_int32(f, binary, len(off) - 1)
    if not binary:
        f.write(b"\n")
    for j in range(len(off) - 1):
        write_diag_gmm(f, am["weights"][off[j]:off[j + 1]], am["means_invvars"][off[j]:off[j + 1]],
                       am["inv_vars"][off[j]:off[j + 1]], binary)


def read_ivector_extractor(s, binary=True):
    """IvectorExtractor::Read (ivector/ivector-extractor.cc:727-748): dict(w [n x S] or empty, w_vec [n],
    M [n x D x S], Sigma_inv [n x D x D] (full), prior_offset), all float64."""
    expect_token(s, binary, "<IvectorExtractor>")
    expect_token(s, binary, "<w>")
    w = read_matrix(s, binary).astype(np.float64)
    expect_token(s, binary, "<w_vec>")
    w_vec = read_vector(s, binary).astype(np.float64)
    expect_token(s, binary, "<M>")
    n = read_int32(s, binary)
    if n <= 0:
        raise ValueError("IvectorExtractor::Read: size > 0")
    M = np.stack([read_matrix(s, binary).astype(np.float64) for _ in range(n)])
    expect_token(s, binary, "<SigmaInv>")
    Si = np.stack([read_sp_matrix(s, binary).astype(np.float64) for _ in range(n)])
    expect_token(s, binary, "<IvectorOffset>")
    off = read_double(s, binary)
    expect_token(s, binary, "</IvectorExtractor>")
    return dict(w=w, w_vec=w_vec, M=M, Sigma_inv=Si, prior_offset=off)


def write_ivector_extractor(f, ext, binary=True):
    """IvectorExtractor::Write (ivector-extractor.cc:706-724)."""
    write_token(f, binary, "<IvectorExtractor>")
    write_token(f, binary, "<w>")
    write_matrix(f, np.asarray(ext["w"], np.float64), binary)
    write_token(f, binary, "<w_vec>")
    write_vector(f, np.asarray(ext["w_vec"], np.float64), binary)
    write_token(f, binary, "<M>")
    M = np.asarray(ext["M"], np.float64)
    write_int32(f, binary, len(M))
    for Mi in M:
        write_matrix(f, Mi, binary)
    write_token(f, binary, "<SigmaInv>")
    for Si in np.asarray(ext["Sigma_inv"], np.float64):
        write_sp_matrix(f, Si, binary)
    write_token(f, binary, "<IvectorOffset>")
    write_double(f, binary, ext["prior_offset"])
    write_token(f, binary, "</IvectorExtractor>")


def read_kaldi_object(path, reader):
    """A Kaldi object file: optional "\\0B" header, then the object (Input::Open + Read)."""
    with open(path, "rb") as f:
        s = Stream(f)
        binary = init_kaldi_input(s)
        return reader(s, binary)


def read_wave(path_or_file):
    """WaveData::Read (feat/wave-reader.cc:105-270) for little-endian RIFF PCM-16 files: returns
    (samp_freq, data [channels x samples] float32 with the int16 values, as the reference keeps them)."""
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, bytes)) else path_or_file
    try:
        b = f.read()
    finally:
        if isinstance(path_or_file, (str, bytes)):
            f.close()
    if b[:4] != b"RIFF":
        raise ValueError("WaveData: expected RIFF, got %r" % b[:4])
    if b[8:12] != b"WAVE" or b[12:16] != b"fmt ":
        raise ValueError("WaveData: expected WAVE and a fmt chunk")
    sub1 = struct.unpack("<I", b[16:20])[0]
    fmt, ch, rate, byte_rate, align, bits = struct.unpack("<HHIIHH", b[20:36])
    if fmt != 1:
        raise ValueError("WaveData: can read only PCM data, format id in file is: %d" % fmt)
    if ch == 0:
        raise ValueError("WaveData: no channels present")
    if bits != 16:
        raise ValueError("WaveData: unsupported bits_per_sample = %d" % bits)
    if byte_rate != rate * bits // 8 * ch or align != ch * bits // 8:
        raise ValueError("WaveData: unexpected byte rate / block align")
    p = 20 + sub1
    while b[p:p + 4] != b"data":       # "fact" / "LIST" chunks between fmt and data are skipped
        if p + 8 > len(b):
            raise ValueError("WaveData: expected data chunk")
        p += 8 + struct.unpack("<I", b[p + 4:p + 8])[0]
    n = struct.unpack("<I", b[p + 4:p + 8])[0]
    raw = np.frombuffer(b[p + 8:p + 8 + n - n % (2 * ch)], "<i2")
    return float(rate), raw.reshape(-1, ch).T.astype(np.float32)


def write_wave(f, samp_freq, data):
    """WaveData::Write (wave-reader.cc:274-360): 16-bit PCM; the samples are truncated to integers
    (static_cast<int32>) and must fit int16 as there."""
    d = np.asarray(data, np.float32)
    if d.ndim == 1:
        d = d[None, :]
    ch, n = d.shape
    q = np.trunc(d.T).astype(np.int64)
    if q.size and (q.min() < -32768 or q.max() > 32767):
        raise ValueError("Wave file is out of range for 16-bit.")
    pcm = q.astype("<i2").tobytes()
    rate = int(samp_freq)
    f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * 2 * ch, 2 * ch, 16) +
            b"data" + struct.pack("<I", len(pcm)) + pcm)


def read_config_file(path):
    """ParseOptions::ReadConfigFile (util/parse-options.cc:451-500): one "--name=value" (or "--flag")
    per line, "#" comments; names keep their dashes -> dict name -> string."""
    out = {}
    with open(path) as f:
        for n, line in enumerate(f, 1):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            if not line.startswith("--"):
                raise ValueError("Reading config file %s: line %d does not look like a line from a Kaldi command-line "
                                 "program's config file: should be of the form --x=y" % (path, n))
            k, _, v = line[2:].partition("=")
            out[k.strip()] = v.strip() if _ else "true"
    return out
