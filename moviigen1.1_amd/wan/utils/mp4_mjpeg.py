"""A self-contained .mp4 writer for the step after the path: uint8 frames -> ISO base media file (ISO/IEC 14496-12) whose one video track carries
JPEG pictures (sample entry `mp4v`, ES descriptor objectTypeIndication 0x6C = "Visual ISO/IEC 10918-1", the registration ffmpeg's own mp4 muxer
uses for mjpeg and every libavformat-based player reads back).

Why it exists: the reference's `cache_video` (wan/utils/utils.py:23-60) hands the frames to imageio / libx264; this image has neither, and through round
4 the frames were left as a `.npy`.  PIL is here: every frame is one baseline JPEG (intra only, every sample a sync sample), the container is written
by hand — `ftyp`, one `mdat` with the pictures back to back, `moov` behind it (one chunk, constant sample duration).  When imageio IS installed
`cache_video` uses it as the reference does; this is the fallback, not a codec claim: H.264 it is not.

`read_mp4_mjpeg` is the matching reader (box walk -> the sample table -> the JPEG byte strings): what the tests use to prove the file is what it says,
and a starting point for anybody who has to get the frames back without a player."""
import io
import struct

import numpy as np

__all__ = ['write_mp4_mjpeg', 'read_mp4_mjpeg']


def _box(kind, *payload):
    body = b''.join(payload)
    return struct.pack('>I4s', 8 + len(body), kind) + body


def _full(kind, version, flags, *payload):
    return _box(kind, struct.pack('>I', (version << 24) | flags), *payload)


def _descr(tag, body):
    """an MPEG-4 systems descriptor: tag, length in the 4-byte expandable form, body (ISO/IEC 14496-1 8.3.3)"""
    n = len(body)
    return bytes([tag, 0x80 | (n >> 21) & 0x7f, 0x80 | (n >> 14) & 0x7f, 0x80 | (n >> 7) & 0x7f, n & 0x7f]) + body


_MATRIX = struct.pack('>9I', 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def write_mp4_mjpeg(path, frames, fps=16, quality=95):
    """frames: uint8 [T, H, W, 3] (RGB) -> `path`.  Returns the number of bytes written."""
    from PIL import Image
    frames = np.asarray(frames)
    if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3 or frames.shape[0] < 1:
        raise ValueError(f'frames must be uint8 [T, H, W, 3] with T >= 1, got {frames.dtype} {frames.shape}')
    T, H, W, _ = frames.shape
    if H >= 1 << 16 or W >= 1 << 16:
        raise ValueError('a sample entry holds 16-bit picture sizes')
    fps = float(fps)
    if not fps > 0:
        raise ValueError('fps must be positive')
    samples = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format='JPEG', quality=int(quality), subsampling=0 if quality >= 95 else 2, optimize=False)
        samples.append(buf.getvalue())
    # media time: 1000 ticks per frame at a timescale of 1000 * fps (exact for every fps with three decimals); movie timescale 1000 (ms)
    delta = 1000
    timescale = int(round(fps * 1000))
    media_duration = T * delta
    movie_duration = int(round(T * 1000.0 / fps))
    payload = sum(len(s) for s in samples)
    ftyp = _box(b'ftyp', b'isom', struct.pack('>I', 0x200), b'isom', b'iso2', b'mp41')
    big = 8 + payload >= 1 << 32
    mdat_head = struct.pack('>I4sQ', 1, b'mdat', 16 + payload) if big else struct.pack('>I4s', 8 + payload, b'mdat')
    bitrate = min(int(payload * 8 * fps / T), 0xffffffff)
    first = len(ftyp) + len(mdat_head)                       # file offset of the first picture = the one chunk
    esds = _full(b'esds', 0, 0, _descr(0x03, struct.pack('>HB', 1, 0) +                                        # ES_ID 1, no optional fields
                                       _descr(0x04, struct.pack('>BB', 0x6C, 0x11) + struct.pack('>I', max(len(s) for s in samples))[1:] +
                                              struct.pack('>II', bitrate, bitrate)) +  # visual stream; buffer size, max and average bit rate
                                       _descr(0x06, b'\x02')))
    entry = _box(b'mp4v', b'\0' * 6, struct.pack('>H', 1),                     # data reference 1
                 b'\0' * 16, struct.pack('>HH', W, H), struct.pack('>II', 0x480000, 0x480000), b'\0' * 4, struct.pack('>H', 1),
                 bytes([5]) + b'mjpeg' + b'\0' * 26, struct.pack('>Hh', 24, -1), esds)
    stbl = _box(b'stbl',
                _full(b'stsd', 0, 0, struct.pack('>I', 1), entry),
                _full(b'stts', 0, 0, struct.pack('>III', 1, T, delta)),
                _full(b'stsc', 0, 0, struct.pack('>IIII', 1, 1, T, 1)),                                           # one chunk with all T samples
                _full(b'stsz', 0, 0, struct.pack('>II', 0, T), b''.join(struct.pack('>I', len(s)) for s in samples)),
                _full(b'co64', 0, 0, struct.pack('>IQ', 1, first)) if first >= 1 << 32 else _full(b'stco', 0, 0, struct.pack('>II', 1, first)))
    minf = _box(b'minf', _full(b'vmhd', 0, 1, b'\0' * 8),
                _box(b'dinf', _full(b'dref', 0, 0, struct.pack('>I', 1), _full(b'url ', 0, 1))),                  # flag 1: the media is in this file
                stbl)
    mdia = _box(b'mdia', _full(b'mdhd', 0, 0, struct.pack('>IIII', 0, 0, timescale, media_duration), struct.pack('>HH', 0x55C4, 0)),   # language "und"
                _full(b'hdlr', 0, 0, b'\0' * 4, b'vide', b'\0' * 12, b'VideoHandler\0'), minf)
    tkhd = _full(b'tkhd', 0, 3, struct.pack('>IIIII', 0, 0, 1, 0, movie_duration), b'\0' * 8, struct.pack('>hhhH', 0, 0, 0, 0), _MATRIX,
                 struct.pack('>II', W << 16, H << 16))
    mvhd = _full(b'mvhd', 0, 0, struct.pack('>IIII', 0, 0, 1000, movie_duration), struct.pack('>IH', 0x10000, 0x100), b'\0' * 10, _MATRIX,
                 b'\0' * 24, struct.pack('>I', 2))                                                              # next track id
    moov = _box(b'moov', mvhd, _box(b'trak', tkhd, mdia))
    with open(path, 'wb') as fh:
        fh.write(ftyp)
        fh.write(mdat_head)
        for s in samples:
            fh.write(s)
        fh.write(moov)
    return len(ftyp) + len(mdat_head) + payload + len(moov)


def _walk(buf, start, end):
    """(kind, payload start, box end) of the boxes in buf[start:end]"""
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack_from('>I4s', buf, pos)
        head = 8
        if size == 1:
            size = struct.unpack_from('>Q', buf, pos + 8)[0]
            head = 16
        elif size == 0:
            size = end - pos
        if size < head or pos + size > end:
            raise ValueError(f'box {kind!r} at {pos}: size {size} does not fit')
        yield kind, pos + head, pos + size
        pos += size


def _find(buf, start, end, *path):
    for kind, a, b in _walk(buf, start, end):
        if kind == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, *path[1:])
    raise ValueError(f'no {path[0]!r} box')


def read_mp4_mjpeg(path, decode=True):
    """-> dict(width, height, fps, codec, object_type, frames) of a file written by write_mp4_mjpeg (one video track, one sample description):
    frames = uint8 [T, H, W, 3] when `decode`, else the list of JPEG byte strings."""
    buf = open(path, 'rb').read()
    a, b = _find(buf, 0, len(buf), b'moov', b'trak', b'mdia')
    ma, _ = _find(buf, a, b, b'mdhd')
    version = buf[ma]
    timescale = struct.unpack_from('>I', buf, ma + (20 if version else 12))[0]
    sa, sb = _find(buf, a, b, b'minf', b'stbl')
    da, db = _find(buf, sa, sb, b'stsd')
    n_desc = struct.unpack_from('>I', buf, da + 4)[0]
    if n_desc != 1:
        raise ValueError('one sample description expected')
    (codec, ea, eb), = list(_walk(buf, da + 8, db))[:1]
    width, height = struct.unpack_from('>HH', buf, ea + 24)
    xa, xb = _find(buf, ea + 78, eb, b'esds')
    # ES_Descriptor -> DecoderConfigDescriptor -> objectTypeIndication (lengths in the expandable form)
    def descr(pos):
        tag = buf[pos]
        pos += 1
        n = 0
        while True:
            c = buf[pos]
            pos += 1
            n = (n << 7) | (c & 0x7f)
            if not c & 0x80:
                break
        return tag, pos, n
    tag, p, _ = descr(xa + 4)
    if tag != 0x03:
        raise ValueError('ES_Descriptor expected')
    tag, p, _ = descr(p + 3)
    if tag != 0x04:
        raise ValueError('DecoderConfigDescriptor expected')
    object_type = buf[p]
    ta, _ = _find(buf, sa, sb, b'stts')
    n_runs = struct.unpack_from('>I', buf, ta + 4)[0]
    runs = [struct.unpack_from('>II', buf, ta + 8 + 8 * i) for i in range(n_runs)]
    za, _ = _find(buf, sa, sb, b'stsz')
    fixed, count = struct.unpack_from('>II', buf, za + 4)
    sizes = [fixed] * count if fixed else list(struct.unpack_from(f'>{count}I', buf, za + 12))
    ca, _ = _find(buf, sa, sb, b'stsc')
    n_sc = struct.unpack_from('>I', buf, ca + 4)[0]
    stsc = [struct.unpack_from('>III', buf, ca + 8 + 12 * i) for i in range(n_sc)]
    try:
        oa, _ = _find(buf, sa, sb, b'stco')
        n_ch = struct.unpack_from('>I', buf, oa + 4)[0]
        chunks = list(struct.unpack_from(f'>{n_ch}I', buf, oa + 8))
    except ValueError:
        oa, _ = _find(buf, sa, sb, b'co64')
        n_ch = struct.unpack_from('>I', buf, oa + 4)[0]
        chunks = list(struct.unpack_from(f'>{n_ch}Q', buf, oa + 8))
    # samples per chunk from the run-length table, then the byte ranges
    per_chunk = []
    for i, (first, n, _idx) in enumerate(stsc):
        last = stsc[i + 1][0] - 1 if i + 1 < len(stsc) else len(chunks)
        per_chunk += [n] * (last - first + 1)
    jpegs, k = [], 0
    for off, n in zip(chunks, per_chunk):
        for _ in range(n):
            jpegs.append(buf[off:off + sizes[k]])
            off += sizes[k]
            k += 1
    if k != count or sum(n for n, _ in runs) != count:
        raise ValueError('sample tables disagree')
    fps = timescale / runs[0][1]
    out = dict(width=width, height=height, fps=fps, codec=codec.decode('latin1'), object_type=object_type)
    if decode:
        from PIL import Image
        out['frames'] = np.stack([np.asarray(Image.open(io.BytesIO(j)).convert('RGB')) for j in jpegs])
    else:
        out['frames'] = jpegs
    return out
