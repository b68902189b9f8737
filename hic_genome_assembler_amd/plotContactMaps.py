"""plotContactMap - the reference's heat-map writer (plotContactMaps.py:15-91), drawn from a device-resident
matrix.

The reference passes the full N x N array to ``numpy.percentile`` (colour limits ``lP`` / ``hP``) and to
xarray's ``pcolormesh`` (one quad per cell; "more than an hour" for large maps by its own warning).  A
32-inch figure at 100 dpi has 3,200 pixels per side, so here

* the colour limits are the exact ``numpy.percentile`` values, selected on the GPU (hicmi_plot_percentiles);
* the image is the block mean of the matrix at figure resolution (hicmi_plot_downsample), shown with
  ``imshow`` in matrix coordinates, so the reference's tick and outline arithmetic carries over unchanged.

Same signature, file names, figure sizes, colour maps ("plasma" + ``reverseColorMap``), Mb tick labels, white
chromosome outlines and titles.  The PNGs are a visual output: they are not compared byte-wise with the
reference's (xarray is not installed here, the reference's plots cannot be produced) - what is tested is the
percentile / downsample arithmetic and that the files are written.

``adjMat`` is a :class:`DeviceImage` (matrix on the GPU) or anything ``numpy.asarray`` accepts (small
host matrices take the same code path with NumPy doing the two reductions).
"""
from __future__ import annotations

import os
import time

import numpy as np

MAX_PIXELS = 3200                  # 32 in x 100 dpi: the largest figure the reference asks for


def plots_enabled(path=True):
    """A plot is written unless its path is empty / False or HICMI_NO_PLOTS is set."""
    return bool(path) and path is not False and not os.environ.get("HICMI_NO_PLOTS")


def figure_pixels(n, wInches, hInches, dpi=100.0):
    return int(max(1, min(n, MAX_PIXELS, max(wInches, hInches) * dpi)))


class DeviceImage:
    """What a plot shows: ``kind`` (0 contacts, 1 distance, 2 similarity - hicmi.h) of the context's matrix,
    restricted / permuted by ``order`` (matrix rows in plot order; None = all rows as stored)."""

    def __init__(self, ctx, kind=0, order=None):
        self.ctx = ctx
        self.kind = int(kind)
        self.order = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
        n = ctx.n if self.order is None else len(self.order)
        self.shape = (n, n)
        self._pct, self._img = {}, {}

    def __len__(self):
        return self.shape[0]

    def percentiles(self, q):
        key = tuple(float(v) for v in q)
        if key not in self._pct:
            self._pct[key] = self.ctx.plot_percentiles(self.kind, self.order, q)
        return self._pct[key]

    def pixels(self, px):
        if px not in self._img:
            self._img[px] = self.ctx.plot_downsample(self.kind, self.order, px)
        return self._img[px]

    def prefetch(self, q, px):
        """Run the two device reductions now (a context is not thread-safe; the drawing that follows is)."""
        self.percentiles(q)
        self.pixels(px)
        return self


class _HostImage:
    def __init__(self, a):
        self.a = np.asarray(a, dtype=np.float64)
        self.shape = self.a.shape

    def __len__(self):
        return self.shape[0]

    def percentiles(self, q):
        return np.percentile(self.a, q)

    def pixels(self, px):
        n = len(self.a)
        if px >= n:
            return self.a
        edges = (np.arange(px + 1, dtype=np.int64) * n) // px
        rows = np.add.reduceat(self.a, edges[:-1], axis=0)
        blocks = np.add.reduceat(rows, edges[:-1], axis=1)
        counts = np.diff(edges)
        return blocks / (counts[:, None] * counts[None, :])


def plotContactMap(adjMat, resolution=100000, tickCount=11, highlightChroms=False, wInches=32, hInches=32, lP=1, hP=98,
                   reverseColorMap='_r', showPlot=False, savePlot=False, title=False, titleSuffix=False):
    """plotContactMaps.py:15-91."""
    from matplotlib.backends.backend_agg import FigureCanvasAgg
    from matplotlib.figure import Figure

    if not isinstance(adjMat, DeviceImage):
        adjMat = _HostImage(adjMat)
    print("- Attempting to plot array / matrix of shape {}".format(adjMat.shape))
    startTime = time.time()
    matrixLength = len(adjMat)
    fig = Figure(figsize=(wInches, hInches))
    FigureCanvasAgg(fig)
    ax = fig.add_subplot(111)
    vmin, vmax = adjMat.percentiles([lP, hP])
    img = adjMat.pixels(figure_pixels(matrixLength, wInches, hInches, fig.get_dpi()))
    # pcolormesh(adjMat[::-1]) puts row 0 at the top: so does imshow with the upper origin
    ax.imshow(img, cmap="plasma" + reverseColorMap, vmin=float(vmin), vmax=float(vmax), origin="upper",
              extent=(0, matrixLength, 0, matrixLength), interpolation="nearest", aspect="auto")
    if highlightChroms is not False:
        prevIndex = 0
        for index in highlightChroms:
            top, bottom = matrixLength - prevIndex, matrixLength - index
            ax.plot([prevIndex, index], [top, top], color='white')
            ax.plot([prevIndex, index], [bottom, bottom], color='white')
            ax.plot([prevIndex, prevIndex], [bottom, top], color='white')
            ax.plot([index, index], [bottom, top], color='white')
            prevIndex = index
        ax.plot([prevIndex, matrixLength], [matrixLength - prevIndex, matrixLength - prevIndex], color='white')
        ax.plot([prevIndex, prevIndex], [0, matrixLength - prevIndex], color='white')
    ax.set_xlim(0, matrixLength)
    ax.set_ylim(0, matrixLength)
    tickDist = matrixLength / tickCount
    xTicks, tickMan = [0], 0
    for _ in range(0, tickCount - 1):
        tickMan += tickDist
        xTicks.append(tickMan)
    xTicks.append(matrixLength)
    ax.set_xticks(xTicks)
    ax.set_xticklabels([str(int((t * resolution) / 1000000)) + " Mb" for t in xTicks], size=18)
    ax.set_xlabel('')
    xTicks.pop(0)
    ax.set_yticks(xTicks)
    ax.set_yticklabels([str(int((t * resolution) / 1000000)) + " Mb" for t in xTicks], size=18)
    ax.set_ylabel('')
    if title is not False:
        if titleSuffix is not False:
            title = title + titleSuffix
        ax.set_title(title, size=25)
    if savePlot is not False:
        fig.savefig(savePlot)
    if showPlot is not False:
        print("- showPlot is ignored (the Agg canvas has no window); the figure is only saved")
    print("Time to rearrange matrix and plot " + str(time.time() - startTime))
