"""Host utilities of the detect path: tile grid, FITS ingest, catalog text.

  generate_tiles      caesar_yolo/utils.py:622-697 (same arguments, same (xmin, xmax_excl, ymin, ymax_excl) tuples,
                      y-outer/x-inner order, None on rejected arguments)
  read_fits_image     value semantics of caesar_yolo/utils.py:193-246 / :340-418 (2-D, or 4-D with degenerate leading axes;
                      the non-finite -> 0 replacement happens on device in cy_mosaic_prepare)
  get_fits_header     caesar_yolo/utils.py:150-165
The reference reads FITS through astropy/fitsio; neither is needed here: a FITS primary HDU is 2880-byte header blocks of
80-character cards followed by big-endian data, which numpy memory-maps directly.
"""
import logging
import os
import numpy as np

logger = logging.getLogger("caesar_yolo_amd")


def generate_tiles(img_xmin, img_xmax, img_ymin, img_ymax, tileSizeX, tileSizeY, gridStepSizeX, gridStepSizeY):
    if img_xmax <= img_xmin:
        logger.error("xmax must be > xmin!")
        return None
    if img_ymax <= img_ymin:
        logger.error("ymax must be > ymin!")
        return None
    if tileSizeX <= 0 or tileSizeY <= 0:
        logger.error("Invalid box size given!")
        return None
    if gridStepSizeX <= 0 or gridStepSizeY <= 0 or gridStepSizeX > 1 or gridStepSizeY > 1:
        logger.error("Invalid grid step size given (null or negative)!")
        return None
    Nx, Ny = img_xmax - img_xmin + 1, img_ymax - img_ymin + 1
    if tileSizeX > Nx or tileSizeY > Ny:
        logger.warning("Invalid box size given (too small or larger than image size)!")
        return None
    stepX, stepY = int(np.round(gridStepSizeX * tileSizeX)), int(np.round(gridStepSizeY * tileSizeY))

    def starts(n, size, step):
        out, i = [], 0
        while i < n:
            out.append((i, i + min(size, n - i)))
            i += step
        return out
    xs, ys = starts(Nx, tileSizeX, stepX), starts(Ny, tileSizeY, stepY)
    return [(img_xmin + x0, img_xmin + x1, img_ymin + y0, img_ymin + y1) for (y0, y1) in ys for (x0, x1) in xs]


# --------------------------------------------------------------------------------------------- FITS
def _parse_card_value(raw):
    v = raw.split("/")[0].strip() if not raw.strip().startswith("'") else raw
    s = v.strip()
    if s.startswith("'"):
        end = s.find("'", 1)
        while end != -1 and end + 1 < len(s) and s[end + 1] == "'":
            end = s.find("'", end + 2)
        return s[1:end].replace("''", "'").rstrip()
    if s in ("T", "F"):
        return s == "T"
    try:
        return int(s)
    except ValueError:
        try:
            return float(s.replace("D", "E"))
        except ValueError:
            return s


def get_fits_header(filename):
    """-> (dict of keyword -> value, data offset in bytes) for the primary HDU, or None."""
    try:
        hdr, off = {}, 0
        with open(filename, "rb") as fp:
            done = False
            while not done:
                block = fp.read(2880)
                if len(block) < 2880:
                    raise ValueError("truncated FITS header")
                off += 2880
                for i in range(0, 2880, 80):
                    card = block[i:i + 80].decode("ascii", "replace")
                    key = card[:8].strip()
                    if key == "END":
                        done = True
                        break
                    if card[8:10] == "= ":
                        hdr[key] = _parse_card_value(card[10:])
        if not hdr.get("SIMPLE", False):
            raise ValueError("not a standard FITS file")
        return hdr, off
    except Exception as ex:
        logger.error("Cannot read image file: %s (%s)" % (filename, ex))
        return None


def read_fits_image(filename):
    """Primary-HDU image as a 2-D array in FILE byte order (big-endian), plus the header.
    4-D cubes with degenerate leading axes give [0,0,:,:] like the reference (utils.py:207-210, :377-380)."""
    res = get_fits_header(filename)
    if res is None:
        return None
    hdr, off = res
    naxis = hdr.get("NAXIS", 0)
    bitpix = hdr.get("BITPIX")
    dt = {-32: ">f4", -64: ">f8", 16: ">i2", 32: ">i4", 8: "u1", 64: ">i8"}.get(bitpix)
    if dt is None or naxis not in (2, 4):
        logger.error("Invalid/unsupported number of channels found in file %s (nchan=%s)!" % (filename, naxis))
        return None
    nx, ny = hdr["NAXIS1"], hdr["NAXIS2"]
    data = np.memmap(filename, dtype=dt, mode="r", offset=off, shape=(ny, nx))      # plane [0,0] of a cube comes first
    if bitpix != -32 or hdr.get("BSCALE", 1) != 1 or hdr.get("BZERO", 0) != 0:
        data = (np.asarray(data, np.float64) * hdr.get("BSCALE", 1) + hdr.get("BZERO", 0)).astype(">f4")
    return data, hdr


def write_fits_image(filename, data, cards=None, bitpix=-32):
    """Minimal FITS image writer, BITPIX -32 (synthetic mosaics for the benchmark and tests) or -64 (the preprocessed
    float64 image the reference's Analyzer.write_fits saves, caesar_yolo/evaluation.py:550-554)."""
    data = np.asarray(data)
    ny, nx = data.shape
    cards_all = [("SIMPLE", True), ("BITPIX", bitpix), ("NAXIS", 2), ("NAXIS1", nx), ("NAXIS2", ny)] + list(cards or [])
    txt = ""
    for k, v in cards_all:
        if isinstance(v, bool):
            val = "T" if v else "F"
        elif isinstance(v, str):
            val = "'%-8s'" % v
            txt += ("%-8s= %-20s" % (k, val)).ljust(80)
            continue
        else:
            val = repr(v) if isinstance(v, float) else str(v)
        txt += ("%-8s= %20s" % (k, val)).ljust(80)
    txt += "END".ljust(80)
    txt = txt.ljust((len(txt) + 2879) // 2880 * 2880)
    with open(filename, "wb") as fp:
        fp.write(txt.encode("ascii"))
        payload = data.astype(">f8" if bitpix == -64 else ">f4").tobytes()
        fp.write(payload)
        fp.write(b"\0" * ((-len(payload)) % 2880))


# --------------------------------------------------------------------------------------------- DS9 regions
CLASS_COLOR_MAP_DS9 = {'bkg': "black", 'spurious': "red", 'compact': "blue", 'extended': "green",
                       'extended-multisland': "yellow", 'flagged': "black", 'diffuse': "magenta"}
# (SFinder's map, caesar_yolo/inference.py:334-342: the tiled catalog's regions)
CLASS_COLOR_MAP_DS9_FRAME = {'bkg': "black", 'spurious': "red", 'compact': "blue", 'extended': "green",
                             'extended-multisland': "orange", 'flagged': "magenta"}
# (Analyzer's map, caesar_yolo/evaluation.py:108-115: single frames and per-tile regions)


def ds9_region_lines(objs, color_map=None, merged_tag=True):
    """One `box(...)` line per catalog object (caesar_yolo/inference.py:1214-1263, evaluation.py:487-528): centre
    x1 + dx/2, y1 + dy/2 (0-based pixel, written 1-based as DS9 'image' coordinates), size dx x dy, text = source name,
    tags = class name [+ BORDER if edge] [+ MERGED if merged], colour by class.
    The reference serialises through the third-party `regions` package (absent here): this is the DS9 text format that
    package emits (header, `image` frame, 1-based centres, fixed-point numbers); byte-compatibility is unpinned."""
    cmap = color_map or CLASS_COLOR_MAP_DS9
    lines = []
    for o in objs:
        dx, dy = o['x2'] - o['x1'], o['y2'] - o['y1']
        xc, yc = o['x1'] + 0.5 * dx, o['y1'] + 0.5 * dy
        tags = [o['class_name']]
        if o.get('edge'):
            tags.append('BORDER')
        if merged_tag and o.get('merged'):
            tags.append('MERGED')
        meta = "text={%s} " % o['name'] + " ".join("tag={%s}" % t for t in tags)
        lines.append("box(%.4f,%.4f,%.4f,%.4f,%.8f) # %s color=%s" % (xc + 1, yc + 1, dx, dy, 0.0, meta,
                                                                       cmap.get(o['class_name'], "white")))
    return lines


def write_ds9_regions(filename, objs, color_map=None, merged_tag=True):
    """-> number of regions written; nothing is written (and a warning logged) for an empty list, like the reference."""
    if not objs:
        logger.warning("Region list with detected objects is empty, nothing to be written...")
        return 0
    with open(filename, "w") as fp:
        fp.write("# Region file format: DS9 astropy/regions\nimage\n")
        for ln in ds9_region_lines(objs, color_map, merged_tag):
            fp.write(ln + "\n")
    return len(objs)


def image_id_of(path):
    return os.path.splitext(os.path.basename(os.path.abspath(path)))[0]
