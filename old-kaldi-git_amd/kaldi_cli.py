"""The command-line layer of the reference's binaries, so that a recipe line (`steps/nnet2/decode.sh:130-136`,
`steps/online/nnet2/decode.sh:118-125`) runs unchanged against tools/*.py:

  * ParseOptions              util/parse-options.{h,cc}: `--name=value`, bare `--flag` for bools, `--config=FILE`
                              (repeatable; read first, the command line overrides it), `--help`, `--print-args`, `--verbose`,
                              `_` == `-` and case-insensitive option names, a lone `--`, prefixed option groups
                              (`--ivector-silence-weighting.silence-weight=`), the usage text and its exit codes;
  * classify_rspecifier /     util/kaldi-table.cc:127-300: `ark`, `scp`, `ark,scp:A,B`, the options b t f nf p (writing) and
    classify_wspecifier       b t o no p np s ns cs ncs (reading);
  * classify_rxfilename /     util/kaldi-io.cc:81-160: "-" / "" = standard input / output, "cmd |" an input pipe, "| cmd" an
    classify_wxfilename       output pipe, "file:offset", plain files, and the strings that are refused;
  * Input / Output            util/kaldi-io.cc:250-460 for those five kinds;
  * SequentialTableReader, RandomAccessTableReader, TableWriter   util/kaldi-table-inl.h over kaldi_io's object codecs.

Pipes and the GPU.  A process that has initialised the GPU must not fork + exec (on this pool that takes the machine down;
see tools/README or DESIGN.md "Command line").  Every pipe child is therefore started by a HELPER PROCESS that is forked
by start_pipe_helper() BEFORE anything touches the GPU and never touches it itself: the tool asks it (over a pipe) to run
`/bin/sh -c cmd` with its standard output or input connected to a FIFO, and opens the FIFO's other end.  That also covers
script files whose entries are commands ("utt1 sox a.flac -t wav - |"), which are only met while decoding."""
import atexit
import os
import shlex
import shutil
import signal
import struct
import sys
import tempfile
import threading
import time

import numpy as np

from . import kaldi_io as kio


class KaldiError(RuntimeError):
    """KALDI_ERR: the binaries catch std::exception in main() and return -1 (255)."""


_program = "kaldi-hip"
_verbose = 0


def set_program_name(name):
    global _program
    _program = name


def verbose_level():
    return _verbose


def log(msg, kind="LOG", where="main()"):
    """KALDI_LOG / KALDI_WARN (base/kaldi-error.cc): `LOG (program:function():file:line) message` on stderr."""
    sys.stderr.write("%s (%s:%s) %s\n" % (kind, _program, where, msg))
    sys.stderr.flush()


def warn(msg, where="main()"):
    log(msg, "WARNING", where)


def vlog(level, msg, where="main()"):
    if _verbose >= level:
        log(msg, "VLOG[%d]" % level, where)


# ---------------------------------------------------------------- ParseOptions
class ParseOptions:
    """util/parse-options.{h,cc}.  register(name, default, doc[, type]) returns nothing; after read(argv) the values are
    attributes of `.values` (a dict keyed by the registered name) and positional arguments are get_arg(1..num_args())."""

    def __init__(self, usage, prefix="", other=None):
        self.usage = usage
        self.prefix = prefix
        self.other = other
        self.values = {}
        self._opts = {}        # normalised name -> (name, type, doc text, is_standard)
        self._order = []
        self.positional = []
        self.argv = None
        if other is None:
            self._register("config", "", "Configuration file to read (this option may be repeated)", str, True)
            self._register("print-args", True, "Print the command line arguments (to stderr)", bool, True)
            self._register("help", False, "Print out usage message", bool, True)
            self._register("verbose", 0, "Verbose level (higher->more logging)", int, True)

    @staticmethod
    def normalize(name):            # NormalizeArgName :497-510
        return name.replace("_", "-").lower()

    def register(self, name, default, doc, typ=None):
        if self.other is not None:      # ParseOptions(prefix, other): the option becomes prefix.name in the other parser
            self.other.register(self.prefix + "." + name, default, doc, typ)
            return
        self._register(name, default, doc, typ, False)

    def _register(self, name, default, doc, typ, standard):
        typ = typ or type(default)
        idx = self.normalize(name)
        if idx in self._opts:
            warn("Registering option twice, ignoring second time: " + name, "RegisterCommon()")
            return
        if typ is bool:
            text = "%s (bool, default = %s)" % (doc, "true" if default else "false")
        elif typ is int:
            text = "%s (int, default = %d)" % (doc, default)
        elif typ is float:
            text = "%s (float, default = %s)" % (doc, _cxx_float(default))
        else:
            text = '%s (string, default = "%s")' % (doc, default)
        self._opts[idx] = (name, typ, text, standard)
        self.values[idx] = default

    def __getitem__(self, name):
        return self.values[self.normalize(name)]

    # ---- Read :313-400
    def read(self, argv):
        self.argv = list(argv)
        global _verbose
        if argv:
            set_program_name(os.path.basename(argv[0]))
        i = 1
        for a in argv[1:]:      # first pass: config files, --help
            if a.startswith("--"):
                if a == "--":
                    break
                key, value, _ = self._split(a)
                if key == "config":
                    self.read_config_file(value.strip())
                if key == "help":
                    self.print_usage()
                    sys.exit(0)
        double_dash_seen = False
        while i < len(argv):
            a = argv[i]
            if a.startswith("--"):
                if a == "--":
                    i += 1
                    double_dash_seen = True
                    break
                key, value, has_eq = self._split(a)
                if not self._set(key, value.strip(), has_eq):
                    self.print_usage(True)
                    raise KaldiError("Invalid option " + a)
                i += 1
            else:
                break
        for a in argv[i:]:
            if a == "--" and not double_dash_seen:
                double_dash_seen = True
            else:
                self.positional.append(a)
        _verbose = int(self.values["verbose"])
        if self.values["print-args"]:
            sys.stderr.write(" ".join(escape(a) for a in argv) + " \n")
            sys.stderr.flush()
        return i

    def _split(self, arg):            # SplitLongArg :471-494
        body = arg[2:]
        if "=" not in body:
            return self.normalize(body), "", False
        key, _, value = body.partition("=")
        if key == "":
            self.print_usage(True)
            raise KaldiError("Invalid option (no key): " + arg)
        return self.normalize(key), value, True

    def _set(self, key, value, has_eq):   # SetOption :514-539
        if key not in self._opts:
            return False
        name, typ, _, _ = self._opts[key]
        if typ is bool:
            if has_eq and value == "":
                raise KaldiError("Invalid option --%s=" % key)
            v = value.lower()
            if v in ("true", "t", "1", ""):
                self.values[key] = True
            elif v in ("false", "f", "0"):
                self.values[key] = False
            else:
                self.print_usage(True)
                raise KaldiError("Invalid format for boolean argument [expected true or false]: " + value)
        elif typ is int:
            try:
                self.values[key] = int(value, 0)      # strtol(..., 0): decimal, 0x hexadecimal, 0 octal
            except ValueError:
                try:
                    self.values[key] = int(_leading_number(value, True), 0)
                except ValueError:
                    self.print_usage(True)
                    raise KaldiError('Invalid integer option "%s"' % value)
        elif typ is float:
            try:
                self.values[key] = float(_leading_number(value, False))
            except ValueError:
                self.print_usage(True)
                raise KaldiError('Invalid floating-point option "%s"' % value)
        else:
            if not has_eq:
                raise KaldiError("Invalid option --" + key)
            self.values[key] = value
        return True

    def read_config_file(self, filename):   # :437-468
        try:
            f = open(filename, "r")
        except OSError:
            raise KaldiError("Cannot open config file: " + filename)
        with f:
            for n, line in enumerate(f, 1):
                line = line.split("#", 1)[0].strip()
                if not line:
                    continue
                if not line.startswith("--"):
                    raise KaldiError("Reading config file %s: line %d does not look like a line from a Kaldi command-line "
                                     "program's config file: should be of the form --x=y.  Note: config files intended to be "
                                     "sourced by shell scripts lack the '--'." % (filename, n))
                key, value, has_eq = self._split(line)
                if not self._set(key, value.strip(), has_eq):
                    self.print_usage(True)
                    raise KaldiError("Invalid option %s in config file %s" % (line, filename))

    def print_usage(self, print_command_line=False):   # :403-436
        out = ["", self.usage]
        app = [v for k, v in sorted(self._opts.items()) if not v[3]]
        if app:
            out.append("Options:")
            out += ["  --%-25s : %s" % (name, text) for name, _, text, _ in app]
            out.append("")
        out.append("Standard options:")
        out += ["  --%-25s : %s" % (name, text) for name, _, text, std in (v for k, v in sorted(self._opts.items())) if std]
        out.append("")
        if print_command_line and self.argv is not None:
            out.append("Command line was: " + " ".join(escape(a) for a in self.argv) + " ")
        sys.stderr.write("\n".join(out) + "\n")
        sys.stderr.flush()

    def num_args(self):
        return len(self.positional)

    def get_arg(self, i):
        if i < 1 or i > len(self.positional):
            raise KaldiError("ParseOptions::GetArg, invalid index %d" % i)
        return self.positional[i - 1]

    def get_opt_arg(self, i):
        return self.positional[i - 1] if i <= len(self.positional) else ""


def _cxx_float(x):
    """operator<< of a float at the stream's default precision (6 significant digits)."""
    return "%g" % x


def _leading_number(s, integer):
    """strtol / strtod accept a numeric prefix ("12abc" -> 12); an empty prefix is the error."""
    import re
    m = re.match(r"\s*[+-]?(0[xX][0-9a-fA-F]+|\d+)" if integer else r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf|nan)", s, re.I)
    if not m:
        raise ValueError(s)
    return m.group(0)


def escape(s):
    """ParseOptions::Escape :262-311: quote what a shell would split or expand."""
    ok = "[]~#^_-+=:.,/"
    if s and all(c.isalnum() or c in ok for c in s):
        return s
    if "'" not in s:
        return "'" + s + "'"
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"').replace("$", "\\$").replace("`", "\\`") + '"'


# ---------------------------------------------------------------- specifier classification
def classify_wspecifier(wspecifier):
    """ClassifyWspecifier kaldi-table.cc:127-213 -> (kind, archive_wxfilename, script_wxfilename, opts); kind in
    {"ark", "scp", "both", None}; opts = dict(binary, flush, permissive)."""
    opts = dict(binary=True, flush=False, permissive=False)
    if ":" not in wspecifier or wspecifier[-1:].isspace():
        return None, "", "", opts
    before, after = wspecifier.split(":", 1)
    kind = None
    for c in _split_opts(before):
        if c == "b":
            opts["binary"] = True
        elif c == "f":
            opts["flush"] = True
        elif c == "nf":
            opts["flush"] = False
        elif c == "t":
            opts["binary"] = False
        elif c == "p":
            opts["permissive"] = True
        elif c == "ark":
            if kind is None:
                kind = "ark"
            else:
                return None, "", "", opts
        elif c == "scp":
            if kind is None:
                kind = "scp"
            elif kind == "ark":
                kind = "both"
            else:
                return None, "", "", opts
        else:
            return None, "", "", opts
    if kind == "ark":
        return kind, after, "", opts
    if kind == "scp":
        return kind, "", after, opts
    if kind == "both":
        if "," not in after:
            return None, "", "", opts
        a, s = after.split(",", 1)
        return kind, a, s, opts
    return None, "", "", opts


def classify_rspecifier(rspecifier):
    """ClassifyRspecifier kaldi-table.cc:217-296 -> (kind, rxfilename, opts); kind in {"ark", "scp", None}."""
    opts = dict(once=False, sorted=False, called_sorted=False, permissive=False)
    if ":" not in rspecifier or rspecifier[-1:].isspace():
        return None, "", opts
    before, after = rspecifier.split(":", 1)
    kind = None
    for c in _split_opts(before):
        if c in ("b", "t"):
            pass
        elif c in ("o", "no"):
            opts["once"] = c == "o"
        elif c in ("p", "np"):
            opts["permissive"] = c == "p"
        elif c in ("s", "ns"):
            opts["sorted"] = c == "s"
        elif c in ("cs", "ncs"):
            opts["called_sorted"] = c == "cs"
        elif c in ("ark", "scp"):
            if kind is None:
                kind = c
            else:
                return None, "", opts
        else:
            return None, "", opts
    return (kind, after, opts) if kind else (None, "", opts)


def _split_opts(s):
    """SplitStringToVector(before_colon, ", ", false): split on ',' and ' ', keeping empty strings."""
    out, cur = [], ""
    for ch in s:
        if ch in ", ":
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def classify_wxfilename(filename):
    """ClassifyWxfilename kaldi-io.cc:81-124 -> "stdout" | "pipe" | "file" | None."""
    if filename in ("", "-"):
        return "stdout"
    if filename[0] == "|":
        return "pipe"
    if filename[0].isspace() or filename[-1].isspace():
        return None
    if filename[0] in "tb" and filename[1:2] == ",":
        return None
    last = filename[-1]
    if last == "|":
        return None
    if last.isdigit():
        d = len(filename) - 1
        while d > 0 and filename[d].isdigit():
            d -= 1
        return None if filename[d] == ":" else "file"
    if "|" in filename:
        warn("Trying to classify wxfilename with pipe symbol in the wrong place (pipe without | at the beginning?): " + filename,
             "ClassifyWxfilename()")
        return None
    return "file"


def classify_rxfilename(filename):
    """ClassifyRxfilename kaldi-io.cc:127-160 -> "stdin" | "pipe" | "file" | "offset" | None."""
    if filename in ("", "-"):
        return "stdin"
    if filename[0] == "|":
        return None
    if filename[0].isspace() or filename[-1].isspace():
        return None
    if filename[0] in "tb" and filename[1:2] == ",":
        return None
    last = filename[-1]
    if last == "|":
        return "pipe"
    if last.isdigit():
        d = len(filename) - 1
        while d > 0 and filename[d].isdigit():
            d -= 1
        return "offset" if filename[d] == ":" else "file"
    if "|" in filename:
        warn("Trying to classify rxfilename with pipe symbol in the wrong place (pipe without | at the end?): " + filename,
             "ClassifyRxfilename()")
        return None
    return "file"


# ---------------------------------------------------------------- the pipe helper
class _PipeHelper:
    """A child process forked before the GPU is initialised; it runs `/bin/sh -c command` on request with the command's
    standard output (input pipes) or standard input (output pipes) attached to a FIFO the requester opens."""

    def __init__(self):
        self.dir = tempfile.mkdtemp(prefix="kaldi_hip_pipes_")
        req_r, req_w = os.pipe()
        rep_r, rep_w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(req_w)
            os.close(rep_r)
            try:
                self._serve(req_r, rep_w)
            finally:
                os._exit(0)
        os.close(req_r)
        os.close(rep_w)
        self.pid, self.req, self.rep = pid, req_w, rep_r
        self.lock = threading.Lock()
        self.n = 0

    @staticmethod
    def _read_exact(fd, n):
        buf = b""
        while len(buf) < n:
            chunk = os.read(fd, n - len(buf))
            if not chunk:
                return None
            buf += chunk
        return buf

    _RUNNING = -0x7fffffff   # reply to a poll of a child that has not exited yet

    def _serve(self, req, rep):
        import subprocess
        signal.signal(signal.SIGINT, signal.SIG_IGN)
        children = {}
        while True:
            head = self._read_exact(req, 9)
            if head is None:
                break
            op, ident, n = struct.unpack("<cII", head)
            body = self._read_exact(req, n) if n else b""
            if op in (b"r", b"w"):
                fifo, cmd = body.decode().split("\0", 1)
                # the FIFO is opened by the shell itself, so this process never blocks on it
                redirect = (" > " if op == b"r" else " < ") + shlex.quote(fifo)
                try:
                    p = subprocess.Popen(["/bin/sh", "-c", "( " + cmd + " )" + redirect], stdin=subprocess.DEVNULL if op == b"r" else None)
                    children[ident] = p
                    status = 0
                except OSError as e:      # the requester raises instead of opening a FIFO nobody will ever open
                    status = e.errno or 1
                os.write(rep, struct.pack("<Ii", ident, status))
            elif op == b"p":      # has the child exited?  (never blocks: a slow child must not stall the other requesters)
                # The child stays in the table: subprocess caches its return code, so every later poll - the watchdog's and
                # the close's wait - gets the same status.  (Popping it on the first poll that saw it gone let the watchdog
                # consume the exit status of a fast command, and the pclose that followed reported a successful pipe as failed.)
                p = children.get(ident)
                rc = -1 if p is None else p.poll()
                if rc is None:
                    rc = self._RUNNING
                os.write(rep, struct.pack("<Ii", ident, rc))
            elif op == b"f":      # the requester has the exit status: drop the entry (a scp of pipes opens thousands of them)
                p = children.pop(ident, None)
                if p is not None and p.poll() is None:
                    children[ident] = p     # still running (forgotten before it was waited for): reaped at the end
            elif op == b"q":
                break
        for p in children.values():
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()

    def _poll(self, ident):
        """Exit status of the child, _RUNNING while it runs, None if the helper is gone."""
        with self.lock:
            try:
                os.write(self.req, struct.pack("<cII", b"p", ident, 0))
            except OSError:
                return None
            rep = self._read_exact(self.rep, 8)
        return struct.unpack("<Ii", rep)[1] if rep else None

    def spawn(self, cmd, reading):
        with self.lock:
            self.n += 1
            ident = self.n
            fifo = os.path.join(self.dir, "p%d" % ident)
            os.mkfifo(fifo)
            body = (fifo + "\0" + cmd).encode()
            try:
                os.write(self.req, struct.pack("<cII", b"r" if reading else b"w", ident, len(body)) + body)
                rep = self._read_exact(self.rep, 8)
            except OSError:
                rep = None
        if rep is None or struct.unpack("<Ii", rep)[1] != 0:
            os.unlink(fifo)
            raise KaldiError("pipe %s: cannot start the command (%s)"
                             % (cmd, "the pipe helper process is gone" if rep is None else os.strerror(struct.unpack("<Ii", rep)[1])))
        # The open below returns when the shell has opened its end.  If the child dies before it does, nobody ever will:
        # a watchdog polls the child and, once it has exited with the open still pending, opens the other end itself
        # (O_RDWR never blocks on a FIFO) so that the open returns and the failure is reported.
        state = {"opened": False, "dead": None}
        stop = threading.Event()

        def watchdog():
            while not stop.wait(0.05):
                rc = self._poll(ident)
                if rc is None or rc != self._RUNNING:
                    if not state["opened"]:
                        state["dead"] = -1 if rc is None else rc
                        try:
                            fd = os.open(fifo, os.O_RDWR | os.O_NONBLOCK)
                            stop.wait(0.2)
                            os.close(fd)
                        except OSError:
                            pass
                    return
        th = threading.Thread(target=watchdog, daemon=True)
        th.start()
        try:
            f = open(fifo, "rb" if reading else "wb")
            state["opened"] = True
        finally:
            stop.set()
        th.join()
        try:
            os.unlink(fifo)
        except OSError:
            pass
        if state["dead"] is not None and (not reading or state["dead"] != 0):
            # (a reader whose command exited with status 0 before the open returned has simply produced its output already:
            # the shell held the FIFO open while it ran)
            f.close()
            self._forget(ident)
            raise KaldiError("pipe %s: the command exited with status %d before its pipe was connected" % (cmd, state["dead"]))
        return ident, f

    def _forget(self, ident):
        """No reply: the helper drops the table entry of a child whose status has been delivered."""
        with self.lock:
            try:
                os.write(self.req, struct.pack("<cII", b"f", ident, 0))
            except OSError:
                pass

    def wait(self, ident):
        """pclose(): the child's exit status.  Polls, so that one slow child does not hold the helper (and with it every
        other pipe of the process) for the duration of its exit."""
        delay = 0.002
        while True:
            rc = self._poll(ident)
            if rc is None:
                return -1
            if rc != self._RUNNING:
                self._forget(ident)
                return rc
            time.sleep(delay)
            delay = min(0.05, delay * 1.5)

    def stop(self):
        try:
            os.write(self.req, struct.pack("<cII", b"q", 0, 0))
            os.close(self.req)
            os.waitpid(self.pid, 0)
        except OSError:
            pass
        shutil.rmtree(self.dir, ignore_errors=True)


_helper = None


def start_pipe_helper():
    """Call once, BEFORE the GPU is touched (tools/*.py do it first thing in main())."""
    global _helper
    if _helper is None:
        _helper = _PipeHelper()
        atexit.register(stop_pipe_helper)   # the FIFO directory does not outlive the process, however it ends its main()
    return _helper


def stop_pipe_helper():
    global _helper
    if _helper is not None:
        _helper.stop()
        _helper = None


class _PipeFile:
    """The parent's end of a pipe to / from a command started by the helper; close() = pclose()."""

    def __init__(self, cmd, reading):
        if _helper is None:
            raise KaldiError("pipe %s: start_pipe_helper() has not been called (it must run before the GPU is initialised)" % cmd)
        self.cmd = cmd
        self.ident, self.f = _helper.spawn(cmd, reading)
        self.reading = reading

    def read(self, n=-1):
        return self.f.read(n)

    def write(self, b):
        return self.f.write(b)

    def flush(self):
        self.f.flush()

    def tell(self):
        raise OSError("a pipe has no position")

    def close(self):
        if self.f is None:
            return 0
        try:
            self.f.close()
        except BrokenPipeError:
            pass
        self.f = None
        rc = _helper.wait(self.ident) if _helper is not None else -1
        if rc != 0:
            warn("Pipe %s had nonzero return status %d" % (self.cmd, rc), "Close()")   # kaldi-io.cc:300-305, 440-445
        return rc


def open_input(rxfilename):
    """Input::Open (kaldi-io.cc:660-735) -> (binary file object, kind).  Raises KaldiError when it cannot be opened."""
    kind = classify_rxfilename(rxfilename)
    if kind == "stdin":
        return sys.stdin.buffer, kind
    if kind == "pipe":
        return _PipeFile(rxfilename[:-1], True), kind
    if kind == "file":
        try:
            return open(rxfilename, "rb"), kind
        except OSError as e:
            raise KaldiError("Error opening input stream %s: %s" % (rxfilename, e))
    if kind == "offset":
        path, _, off = rxfilename.rpartition(":")
        try:
            f = open(path, "rb")
            f.seek(int(off))
            return f, kind
        except OSError as e:
            raise KaldiError("Error opening input stream %s: %s" % (rxfilename, e))
    raise KaldiError("Invalid input filename format " + escape(rxfilename))


def open_output(wxfilename):
    """Output::Open (kaldi-io.cc:585-640) -> (binary file object, kind)."""
    kind = classify_wxfilename(wxfilename)
    if kind == "stdout":
        return sys.stdout.buffer, kind
    if kind == "pipe":
        return _PipeFile(wxfilename[1:], False), kind
    if kind == "file":
        try:
            return open(wxfilename, "wb"), kind
        except OSError as e:
            raise KaldiError("Error opening output stream %s: %s" % (wxfilename, e))
    raise KaldiError("Invalid output filename format " + escape(wxfilename))


def _close(f, kind):
    if kind in ("stdin", "stdout"):
        if kind == "stdout":
            f.flush()
        return 0
    r = f.close()
    return r if isinstance(r, int) else 0


def read_kaldi_object(rxfilename, reader):
    """`Input ki(rxfilename, &binary); obj.Read(ki.Stream(), binary)` for a model file / FST given as any rxfilename."""
    f, kind = open_input(rxfilename)
    try:
        s = kio.Stream(f)
        binary = kio.init_kaldi_input(s)
        return reader(s, binary)
    finally:
        _close(f, kind)


# ---------------------------------------------------------------- tables
class SequentialTableReader:
    """SequentialTableReader<Holder> (kaldi-table-inl.h:63-520): iterate (key, object) of an rspecifier.  `kind` names the
    Holder ("matrix", "vector", "int32_vector", "lattice", "compact_lattice", "wave", "token_vector", "fst" = fst::VectorFstHolder)."""

    def __init__(self, rspecifier, kind="matrix"):
        self.kind = kind
        self.rspecifier = rspecifier
        self.rkind, self.rx, self.opts = classify_rspecifier(rspecifier)
        if self.rkind is None:
            raise KaldiError("Invalid rspecifier " + escape(rspecifier))
        self.f, self.fkind = open_input(self.rx)
        self.rc = 0

    def __iter__(self):
        try:
            if self.rkind == "ark":
                s = kio.Stream(self.f)
                while True:
                    s.skip_ws()
                    if s.eof():
                        break
                    key = kio.read_token(s, False)
                    if self.kind == "fst":        # fst::VectorFstHolder has no binary-mode header (fstext-utils.h:403-406)
                        yield key, kio.read_fst_holder(s)
                        continue
                    binary = kio.init_kaldi_input(s)
                    yield key, _read_holder(s, binary, self.kind)
            else:
                for line in _text_lines(self.f):
                    parts = line.split(None, 1)
                    if len(parts) != 2:
                        raise KaldiError("Invalid line in script file %s: %r" % (self.rx, line))
                    key, rx = parts[0], parts[1].strip()
                    try:
                        g, gkind = open_input(rx)
                    except KaldiError:
                        if self.opts["permissive"]:       # "p": skip what cannot be read
                            warn("Failed to open %s (key %s): skipped [permissive]" % (rx, key), "SequentialTableReader")
                            continue
                        raise
                    try:
                        s = kio.Stream(g)
                        if self.kind == "fst":
                            obj = kio.read_fst_holder(s)
                        else:
                            binary = kio.init_kaldi_input(s)
                            obj = _read_holder(s, binary, self.kind)
                    finally:
                        _close(g, gkind)
                    yield key, obj
        finally:
            self.close()

    def close(self):
        if self.f is not None:
            self.rc = _close(self.f, self.fkind)
            self.f = None
        return self.rc == 0


class RandomAccessTableReader:
    """RandomAccessTableReader<Holder> (kaldi-table-inl.h:1100-2200).  An archive is read into memory at the first lookup
    (the reference caches as it scans; `s` / `cs` let it free what it has passed - an optimisation, not a semantic); a
    script file is indexed by key and its entries are read on demand."""

    def __init__(self, rspecifier, kind="matrix"):
        self.kind = kind
        self.rkind, self.rx, self.opts = classify_rspecifier(rspecifier)
        if self.rkind is None:
            raise KaldiError("Invalid rspecifier " + escape(rspecifier))
        self._table = None

    def _load(self):
        if self._table is not None:
            return
        self._table = {}
        if self.rkind == "ark":
            for k, v in SequentialTableReader("ark:" + self.rx, self.kind):
                self._table[k] = v
        else:
            f, fk = open_input(self.rx)
            try:
                for line in _text_lines(f):
                    key, rx = line.split(None, 1)
                    self._table[key] = ("rx", rx.strip())
            finally:
                _close(f, fk)

    def has_key(self, key):
        self._load()
        return key in self._table

    def value(self, key):
        self._load()
        if key not in self._table:
            raise KaldiError("Value() called but no such key %s in table" % key)
        v = self._table[key]
        if isinstance(v, tuple) and len(v) == 2 and v[0] == "rx":
            g, gk = open_input(v[1])
            try:
                s = kio.Stream(g)
                binary = kio.init_kaldi_input(s)
                return _read_holder(s, binary, self.kind)
            finally:
                _close(g, gk)
        return v


class TableWriter:
    """TableWriter<Holder> (kaldi-table-inl.h:760-1100) for "ark:", "scp:", "ark,scp:" wspecifiers with the b / t / f / p
    options, archives on files, pipes or standard output.  An empty wspecifier gives a writer that is not open (the
    binaries' optional outputs: `Int32VectorWriter words_writer(words_wspecifier)`)."""

    def __init__(self, wspecifier, kind="matrix"):
        self.kind = kind
        self.f = self.scp = None
        self.wkind = None
        if wspecifier == "":
            return
        self.wkind, self.ark_wx, self.scp_wx, self.opts = classify_wspecifier(wspecifier)
        if self.wkind is None:
            raise KaldiError("Invalid wspecifier " + escape(wspecifier))
        if self.wkind == "scp":
            raise KaldiError("wspecifier %s: writing a script file alone (one file per key) is not supported" % escape(wspecifier))
        self.f, self.fkind = open_output(self.ark_wx)
        if self.wkind == "both":
            if self.fkind != "file":
                raise KaldiError("wspecifier %s: an archive with a script file has to be a real file" % escape(wspecifier))
            self.scp, self.skind = open_output(self.scp_wx)

    def is_open(self):
        return self.f is not None

    def write(self, key, obj):
        if self.f is None:
            return
        if not key or any(c.isspace() for c in key):
            raise KaldiError("Using invalid key %r" % key)
        binary = self.opts["binary"]
        self.f.write(key.encode() + b" ")
        if self.scp is not None:
            self.scp.write(("%s %s:%d\n" % (key, self.ark_wx, self.f.tell())).encode())
        if self.kind == "fst":                # (no header: the holder's binary form is the OpenFst file)
            kio.write_fst_holder(self.f, obj, binary)
            if self.opts["flush"]:
                self.f.flush()
            return
        if binary:
            self.f.write(b"\0B")
        kio._write_object(self.f, binary, self.kind, obj)
        if self.opts["flush"]:
            self.f.flush()

    def close(self):
        ok = True
        if self.f is not None:
            ok = _close(self.f, self.fkind) == 0
            self.f = None
        if self.scp is not None:
            _close(self.scp, self.skind)
            self.scp = None
        return ok


def _text_lines(f):
    data = f.read()
    for line in data.decode().splitlines():
        line = line.strip()
        if line:
            yield line


def read_wave_stream(s):
    """WaveData::Read (feat/wave-reader.cc:105-270) from a stream that may hold more behind the wave (an archive entry) or
    not know its own length (a pipe from sox: RIFF / data sizes of 0 or 0xFFFFFFFF mean "until the end of the stream").
    Returns (samp_freq, data [channels x samples] float32 holding the int16 values)."""
    import struct
    head = s.get(12)
    if head[:4] != b"RIFF":
        raise KaldiError("WaveData: expected RIFF or RIFX, got %r" % head[:4])
    if head[8:12] != b"WAVE":
        raise KaldiError("WaveData: expected WAVE, got %r" % head[8:12])
    fmt = None
    while True:
        cid, size = struct.unpack("<4sI", s.get(8))
        if cid == b"fmt ":
            body = s.get(size)
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            if fmt is None:
                raise KaldiError("WaveData: data chunk before the fmt chunk")
            if size in (0, 0xFFFFFFFF, 0x7FFFFFFF):      # streamed: read to the end
                chunks = []
                while not s.eof():
                    chunks.append(s.get(min(len(s.buf), 1 << 20) or 1))
                raw = b"".join(chunks)
            else:
                raw = s.get(size)
                if size % 2 == 1 and not s.eof() and s.peek(1) == b"\0":
                    s.get(1)
            break
        else:
            s.get(size + (size & 1))                      # "fact", "LIST", ...: skipped
    fmt_id, ch, rate, byte_rate, align, bits = fmt
    if fmt_id != 1:
        raise KaldiError("WaveData: can read only PCM data, format id in file is: %d" % fmt_id)
    if ch == 0:
        raise KaldiError("WaveData: no channels present")
    if bits != 16:
        raise KaldiError("WaveData: unsupported bits_per_sample = %d" % bits)
    raw = raw[:len(raw) - len(raw) % (2 * ch)]
    data = np.frombuffer(raw, "<i2").reshape(-1, ch).T.astype(np.float32)
    return float(rate), np.ascontiguousarray(data)


def _read_holder(s, binary, kind):
    if kind == "wave":
        return read_wave_stream(s)
    if kind == "token_vector":        # TokenVectorHolder: the rest of the line, whitespace separated (text only)
        line = b""
        while True:
            b = s.get(1) if not s.eof() else b""
            if b in (b"", b"\n"):
                break
            line += b
        return line.decode().split()
    return kio._read_object(s, binary, kind)


def read_symbol_table(rxfilename):
    """fst::SymbolTable::ReadText: lines "symbol id" -> {id: symbol}."""
    f, kind = open_input(rxfilename)
    try:
        out = {}
        for line in _text_lines(f):
            sym, idx = line.split()
            out[int(idx)] = sym
        return out
    finally:
        _close(f, kind)
