"""CPU-only tests of the reader side (sdrterm_amd/misc/file_util.py, read_file.py), modelled on the reference's
test/misc/file_util_test.py and read_file_test.py; the header byte strings are that test module's fixtures (data)."""
import pickle
from string import ascii_letters

import numpy as np
import pytest

from sdrterm_amd.misc.file_util import DataType, MappableEnum, checkWavHeader
from sdrterm_amd.misc.read_file import RawChunk, generateDomain, readFile

HEADERS = {
    'uint8': b'RIFF&\x00\x00\x00WAVEfmt \x10\x00\x00\x00\x01\x00\x01\x00@\x1f\x00\x00@\x1f\x00\x00\x01\x00\x08\x00data\x01\x00\x00\x00\n\x00',
    'int16': b'RIFF$\x00\x00\x00WAVEfmt \x10\x00\x00\x00\x01\x00\x01\x00@\x1f\x00\x00\x80>\x00\x00\x02\x00\x10\x00data\x00\x00\x00\x00',
    'int32': b'RIFFH\x00\x00\x00WAVEfmt (\x00\x00\x00\xfe\xff\x01\x00@\x1f\x00\x00\x00}\x00\x00\x04\x00 \x00\x16\x00 \x00\x04\x00\x00\x00\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x008\x9bqfact\x04\x00\x00\x00\x00\x00\x00\x00data\x00\x00\x00\x00',
    'float32': b'RIFF2\x00\x00\x00WAVEfmt \x12\x00\x00\x00\x03\x00\x01\x00@\x1f\x00\x00\x00}\x00\x00\x04\x00 \x00\x00\x00fact\x04\x00\x00\x00\x00\x00\x00\x00data\x00\x00\x00\x00',
    'float64': b'RIFF2\x00\x00\x00WAVEfmt \x12\x00\x00\x00\x03\x00\x01\x00@\x1f\x00\x00\x00\xfa\x00\x00\x08\x00@\x00\x00\x00fact\x04\x00\x00\x00\x00\x00\x00\x00data\x00\x00\x00\x00',
    'bigendian': b'RIFX\x00\x00\x002WAVEfmt \x00\x00\x00\x12\x00\x03\x00\x01\x00\x00\x1f@\x00\x00\xfa\x00\x00\x08\x00@\x00\x00fact\x00\x00\x00\x04\x00\x00\x00\x00data\x00\x00\x00\x00',
    'twochannel': b'RIFF2\x00\x00\x00WAVEfmt \x12\x00\x00\x00\x03\x00\x02\x00@\x1f\x00\x00\x00\xf4\x01\x00\x10\x00@\x00\x00\x00fact\x04\x00\x00\x00\x00\x00\x00\x00data\x00\x00\x00\x00',
}
EXPECT = {'uint8': 'uint8', 'int16': 'int16', 'int32': 'int32', 'float32': 'float32', 'float64': 'float64',
          'bigendian': '>f8', 'twochannel': '<f8'}
BAD = {
    'bad1': b'RIFA2\x00\x00\x00',
    'bad2': b'RIFF2\x00\x00\x00AVI\x00\x00',
    'bad3': b'RIFF2\x00\x00\x00WAVEtmf \x12',
    'bad5': b'RIFFH\x00\x00\x00WAVEfmt (\x00\x00\x00\xfe\xff\x01\x00@\x1f\x00\x00\x00}\x00\x00\x04\x00 \x00\x16\x00 \x00\x04\x00\x00\x00\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x008\x9bf',
    'bad6': b'RIFF$\x00\x00\x00WAVEfmt \x10\x00\x00\x00\x01\x00\x01\x00@\x1f\x00\x00\x80>\x00\x00\x02\x00\x11\x00data',
    'badfloat': b'RIFF2\x00\x00\x00WAVEfmt \x12\x00\x00\x00\x03\x00\x01\x00@\x1f\x00\x00\x00\xfa\x00\x00\x08\x00\x41\x00\x00\x00fact\x04\x00\x00\x00\x00\x00\x00\x00data\x00\x00\x00\x00',
    'alaw': b'RIFF2\x00\x00\x00WAVEfmt \x12\x00\x00\x00\x06\x00\x02\x00@\x1f\x00\x00\x80>\x00\x00\x02\x00\x08\x00\x00\x00fact\x04\x00\x00\x00\x00\x00\x00\x00data\x00\x00\x00\x00',
}


def _file(tmp_path, name, data, suffix='.bin'):
    p = tmp_path / (name + suffix)
    p.write_bytes(data)
    return str(p)


@pytest.mark.parametrize('name', sorted(HEADERS))
def test_wav_headers(tmp_path, name):
    info = checkWavHeader(_file(tmp_path, name, HEADERS[name]), 0, '')
    assert info['bitsPerSample'] == np.dtype(EXPECT[name])
    assert info['sampRate'] == 8000
    assert info['numChannels'] == (2 if name == 'twochannel' else 1)
    assert info['isSocket'] is False
    # the reference's offset: right after the 'data' tag, i.e. AT the chunk-size field (file_util.py:185-194)
    assert info['dataOffset'] == HEADERS[name].index(b'data') + 4
    strict = checkWavHeader(_file(tmp_path, name + 's', HEADERS[name]), 0, '', strictDataOffset=True)
    assert strict['dataOffset'] == info['dataOffset'] + 4


@pytest.mark.parametrize('name', sorted(BAD))
def test_bad_headers(tmp_path, name):
    with pytest.raises(ValueError):
        checkWavHeader(_file(tmp_path, name, BAD[name]), 8000, 'B')


def test_raw_inputs(tmp_path):
    raw = _file(tmp_path, 'raw', b'\x00\x00')
    for f, enc in zip(('localhost:1234', None, raw), ('b', 'B', 'h')):
        info = checkWavHeader(f, 8000, enc)
        assert info['bitsPerSample'] == DataType[enc].value and info['sampRate'] == 8000 and info['dataOffset'] == 0
    assert checkWavHeader('localhost:1234', 8000, 'h')['isSocket'] is True
    assert checkWavHeader('localhost:1234', 8000, 'h')['bitsPerSample'] == np.dtype('>i2')
    with pytest.raises(ValueError):                       # a *.wav that does not start with RIFF (file_util.py:137-139)
        checkWavHeader(_file(tmp_path, 'foo', b'ASDF2\x00\x00\x00WAVEfmt \x12', suffix='.wav'), 8003, 'B')
    for f in ('', b''):
        with pytest.raises(FileNotFoundError):
            checkWavHeader(f, 8000, 'B')
    for ch in sorted(set(ascii_letters) - set('bBhHiIfd')):
        with pytest.raises(KeyError):
            checkWavHeader(None, 8000, ch)
    for fs in (-1, 0, None):
        with pytest.raises(ValueError):
            checkWavHeader(None, fs, 'B')
    with pytest.raises(TypeError):
        checkWavHeader(None, '2', 'B')


def test_enums():
    assert not len(MappableEnum.tuples()) and not len(MappableEnum.dict())
    for member, (k, v) in zip(DataType, DataType.dict().items()):
        assert member.name == k and member.value == v and str(member) == k
    assert [k for k, _ in DataType.tuples()] == list('bBhHiIfd')


def test_generate_domain_and_missing_fs():
    with pytest.raises(ValueError):
        readFile()
    chars = ('B', 'b', 'H', 'h', 'I', 'i', 'L', 'l')
    vals = (0, 255, -128, 127, 0, 65536, -32768, 32767, 0, 4294967295, -2147483648, 2147483647,
            0, 18446744073709551615, -9223372036854775808, 9223372036854775807)
    for c, (a, b) in zip(chars, [(vals[i], vals[i + 1]) for i in range(0, len(vals), 2)]):
        xmin, diff = generateDomain(c)
        assert xmin == a and diff == 1 / (b - a)
    for c in sorted(set(ascii_letters) - set(chars)):
        assert generateDomain(c) is None


class _Proc:
    def __init__(self, alive=True):
        self.exitcode = None if alive else 0
        self.name = 'proc'


class _Q:
    def __init__(self):
        self.items, self.closed = [], False

    def put_nowait(self, x):
        if self.closed:
            raise ValueError('closed')
        self.items.append(x)

    def close(self):
        self.closed = True


class _Flag:
    value = 0


def test_reader_chunks_stale_tail_and_eof(tmp_path):
    """Full reads, then a short one: the reference hands on the whole buffer again, stale tail included
    (read_file.py:119-125); EOF is b''; a consumer whose process ended is dropped."""
    rng = np.random.default_rng(1)
    payload = rng.integers(0, 256, 3 * 4096 + 1000, dtype=np.uint8).tobytes()
    path = _file(tmp_path, 'data', b'HDR!' + payload)
    live, dead = _Q(), _Q()
    readFile(bitsPerSample=np.dtype('<i2'), dataOffset=4, fs=48000, buffers=[live, dead], processes=[_Proc(), _Proc(False)],
             isDead=_Flag(), inFile=path, readSize=4096, correctIq=True, normalize=True, impedance=75)
    assert dead.items == [] and dead.closed                # dropped at the first chunk; the final b'' bounces off the closed Queue
    *chunks, eof = live.items
    assert eof == b'' and live.closed and len(chunks) == 4
    for k in range(3):
        assert chunks[k].data == payload[k * 4096:(k + 1) * 4096]
    assert chunks[3].data == payload[3 * 4096:] + payload[2 * 4096 + 1000:3 * 4096]     # short read + stale tail
    c = chunks[0]
    assert (c.code, c.bigEndian, c.normalize, c.correctIq, c.impedance, c.size) == ('h', False, True, True, 75, 1024)
    d = pickle.loads(pickle.dumps(c))
    assert d.data == c.data and d.code == 'h' and d.size == 1024
    # swapEndianness flips the byte order; float input is never normalised (generateDomain is None for 'f')
    q = _Q()
    readFile(bitsPerSample=np.dtype('<f4'), fs=48000, buffers=[q], processes=[_Proc()], isDead=_Flag(), inFile=path,
             readSize=4096, swapEndianness=True, normalize=True)
    assert isinstance(q.items[0], RawChunk) and q.items[0].bigEndian and not q.items[0].normalize and q.items[0].code == 'f'
