"""Drives the REFERENCE's own GUI glue headless, for oracle/gen_golden.py (build container only; TEST INFRASTRUCTURE).

The reference's glue between its L2 functions lives in Qt / vispy classes (dropout_healer_gui.Canvas.resample_files and
.on_mouse_release, util.markers.MasterSpeedLine / MasterRegLine / TraceLine / RegLine / DropoutSample,
pyrespeeder_gui.Canvas.get_speed_curve).  None of PyQt5, vispy, soundfile, sounddevice, matplotlib, numba is installed, so
an import hook hands out throw-away stand-ins for exactly those packages (any attribute is a class that can be subclassed,
called and ignored; numba.jit is the identity) -- the reference's modules then import unchanged and their methods are called
as plain functions on minimal stand-in canvases that carry only the attributes the methods read.  Nothing of the reference
is copied: only arrays it computed are stored by gen_golden.py.
"""
import importlib.abc
import importlib.machinery
import logging
import sys
import types

import numpy as np

STUBBED = {"PyQt5", "vispy", "soundfile", "sounddevice", "matplotlib", "pyfftw", "numba", "resampy", "librosa", "OpenGL"}


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        sub = _Meta(name, (_Anything,), {})
        setattr(cls, name, sub)
        return sub

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls

    def __iter__(cls):
        return iter(())


class _Anything(metaclass=_Meta):
    """An object that accepts any construction, call and attribute access (a stand-in for Qt widgets and vispy visuals)."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self

    def __bool__(self):
        return False

    def __len__(self):
        return 0


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        cls = _Meta(name, (_Anything,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == "numba":
            def jit(*a, **k):
                if len(a) == 1 and callable(a[0]) and not k:
                    return a[0]
                return lambda fn: fn
            module.jit = jit


_installed = False


def import_reference_gui(ref):
    """-> (dropout_healer_gui, pyrespeeder_gui, util.markers, util.wow_detection) of the reference checkout at `ref`."""
    global _installed
    if not _installed:
        sys.meta_path.insert(0, _Finder())
        _installed = True
    if ref not in sys.path:
        sys.path.insert(0, ref)
    logging.disable(logging.CRITICAL)
    import dropout_healer_gui
    import pyrespeeder_gui
    from util import markers, wow_detection
    return dropout_healer_gui, pyrespeeder_gui, markers, wow_detection


class NS:
    """attribute bag"""
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _bind(obj, cls, *names):
    for n in names:
        setattr(obj, n, types.MethodType(getattr(cls, n), obj))


def fake_canvas(sr, fft_size, hop, duration=0.0):
    """What BaseMarker / BaseLine and the Canvas methods below read of a canvas (util/markers.py:28-38, 566-598)."""
    c = NS(markers=[], views=[], spectra=[_Anything()], sr=sr, fft_size=fft_size, hop=hop, duration=duration,
           speed_view=_Anything(), filenames=["a.wav", "a.wav"],
           props=NS(dropout_widget=NS(surrounding=0.5, width=20, sensitivity=5),
                    files_widget=NS(files=[NS(channel_widget=NS(channels=[0]))]),
                    output_widget=NS(bump_index=lambda: None, suffix="")))
    return c


# ---------------------------------------------------------------------------------------- dropout healer (config 4)
def heal_through_reference(D, markers_mod, signal2d, sr, marks, fft_size, hop, channels=(0,)):
    """dropout_healer_gui.Canvas.resample_files (dropout_healer_gui.py:111-166) itself, on `signal2d` (n, ch) with the
    saved marker tuples `marks` = DropoutSample.to_cfg() rows; returns the array it hands to io_ops.write_file."""
    c = fake_canvas(sr, fft_size, hop)
    c.props.files_widget.files[0].channel_widget.channels = list(channels)
    _bind(c, D.Canvas, "time_2_frame", "frame_2_time", "freq_2_bin")
    c.markers = []
    for m in marks:
        markers_mod.DropoutSample.from_cfg(c, *m).initialize()          # BaseMarker.initialize appends to canvas.markers
    written = {}
    old_r, old_w = D.io_ops.read_file, D.io_ops.write_file
    D.io_ops.read_file = lambda path: (signal2d, sr, signal2d.shape[1])
    D.io_ops.write_file = lambda path, data, sr_, ch, suffix="": written.update(data=np.array(data), suffix=suffix)
    try:
        D.Canvas.resample_files(c, ["a.wav"])
    finally:
        D.io_ops.read_file, D.io_ops.write_file = old_r, old_w
    return written["data"], written["suffix"]


def detect_through_reference(D, markers_mod, magnitudes, sr, fft_size, hop, t_0, t_1, f_lower, f_upper, width_ms, sensitivity):
    """The batch-detection branch of dropout_healer_gui.Canvas.on_mouse_release (dropout_healer_gui.py:168-242) itself: an
    Alt-drag from (t_0, f_lower) to (t_1, f_upper) over the cached magnitude spectrogram.  Returns the corner pairs of the
    DropoutSample markers it pushes onto the undo stack."""
    c = fake_canvas(sr, fft_size, hop)
    c.props.dropout_widget.width = width_ms
    c.props.dropout_widget.sensitivity = sensitivity
    _bind(c, D.Canvas, "time_2_frame", "frame_2_time", "freq_2_bin")
    c.px_to_spectrum = lambda click: click
    spec = NS(sr=sr, key="k", fft_storage={"k": magnitudes})
    from util import spectrum as ref_spectrum
    spec.get_times_freqs = types.MethodType(ref_spectrum.Spectrum.get_times_freqs, spec)
    c.spectra = [spec, spec]
    pushed = []
    c.props.undo_stack = NS(push=lambda action: pushed.extend(action.traces))
    a, b = (t_0, f_lower), (t_1, f_upper)
    event = NS(trail=lambda: [a], pos=b, button=1, modifiers=("Alt",))
    # DropoutSample's visuals want spectra[-1].mel_transform: any object
    spec.mel_transform = _Anything()
    D.Canvas.on_mouse_release(c, event)
    return [((m.a[0], m.a[1]), (m.b[0], m.b[1])) for m in pushed]


# ---------------------------------------------------------------------------------------- pyrespeeder (config 3, .spd)
def respeeder_canvas(P, markers_mod, sr, hop, duration, fft_size=1024):
    """A stand-in pyrespeeder canvas with the reference's own MasterSpeedLine / MasterRegLine on it."""
    c = fake_canvas(sr, fft_size, hop, duration)
    c.master_speed = markers_mod.MasterSpeedLine(c)
    c.master_reg_speed = markers_mod.MasterRegLine(c, (0, 0, 1, .5))
    type(c).lines = property(lambda self: [m for m in self.markers if isinstance(m, markers_mod.TraceLine)])
    type(c).regs = property(lambda self: [m for m in self.markers if isinstance(m, markers_mod.RegLine)])
    _bind(c, P.Canvas, "get_speed_curve", "update_lines")
    return c


def speed_curve_through_reference(P, markers_mod, sr, hop, duration, lines, regs, bands=(0, 20)):
    """pyrespeeder_gui.Canvas.get_speed_curve (pyrespeeder_gui.py:133-140) after update_lines (:159-161) on a canvas that
    holds TraceLine.from_cfg(*row) for `lines` = [[times, freqs, offset], ...] and RegLine.from_cfg(*row) for `regs` =
    [[t0, t1, amplitude, omega, phase, offset], ...] -- exactly what util/widgets.py:1247-1262 builds from a .spd file.
    Returns (curve, master_speed.data, master_reg_speed.data)."""
    class Canvas(NS):
        pass
    c0 = respeeder_canvas(P, markers_mod, sr, hop, duration)
    c = Canvas(**c0.__dict__)
    Canvas.lines = property(lambda self: [m for m in self.markers if isinstance(m, markers_mod.TraceLine)])
    Canvas.regs = property(lambda self: [m for m in self.markers if isinstance(m, markers_mod.RegLine)])
    c.master_speed.vispy_canvas = c
    c.master_reg_speed.vispy_canvas = c
    _bind(c, P.Canvas, "get_speed_curve", "update_lines")
    for row in lines:
        markers_mod.TraceLine.from_cfg(c, *row).initialize()
    for row in regs:
        markers_mod.RegLine.from_cfg(c, *row).initialize()
    c.master_speed.bands = bands
    c.update_lines()
    return np.array(c.get_speed_curve()), np.array(c.master_speed.data), np.array(c.master_reg_speed.data)


# ---------------------------------------------------------------------------------------- heuristic dropout repair (8f-3)
def heuristic_through_reference(ref, signal2d, sr, fft_size, hop, max_width=0.02, max_slope=0.5, num_bands=3, bottom_freedom=2,
                                f_upper=12000, f_lower=3000):
    """dropouts_gui.MainWindow.process_heuristic (dropouts_gui.py:241-323) itself on `signal2d` (n, ch) float32: the method is
    called as a plain function on a stand-in window that carries the DropoutWidget values (util/widgets.py:832-889 defaults)
    and one file name; io_ops.read_file / write_file are swapped for in-memory ones.  Returns the array it writes."""
    global _installed
    if not _installed:
        sys.meta_path.insert(0, _Finder())
        _installed = True
    if ref not in sys.path:
        sys.path.insert(0, ref)
    logging.disable(logging.CRITICAL)
    import dropouts_gui as G
    win = NS(dropout_widget=NS(max_width=max_width, max_slope=max_slope, num_bands=num_bands, bottom_freedom=bottom_freedom,
                               f_upper=f_upper, f_lower=f_lower),
             file_names=["a.wav"], names_to_full_paths={"a.wav": "a.wav"})
    written = {}
    old_r, old_w = G.io_ops.read_file, G.io_ops.write_file
    G.io_ops.read_file = lambda path: (np.array(signal2d, dtype=np.float32), sr, signal2d.shape[1])
    G.io_ops.write_file = lambda path, data, sr_, ch, suffix="_out": written.update(data=np.array(data), suffix=suffix)
    try:
        with np.errstate(all="ignore"):
            G.MainWindow.process_heuristic(win, fft_size, hop)
    finally:
        G.io_ops.read_file, G.io_ops.write_file = old_r, old_w
    return written["data"]
