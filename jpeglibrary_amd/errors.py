"""Exception classes mirroring the ones the reference throws (System.IO.InvalidDataException etc.)."""
from . import _capi


class JpegError(Exception):
    status = None


class InvalidDataException(JpegError):
    status = _capi.ERR_INVALID_DATA


class InvalidOperationException(JpegError):
    status = _capi.ERR_INVALID_OPERATION


class NotSupportedException(JpegError):
    status = _capi.ERR_NOT_SUPPORTED


class ArgumentException(JpegError):
    status = _capi.ERR_ARGUMENT


class DeviceError(JpegError):
    status = _capi.ERR_DEVICE


class NoDeviceError(DeviceError):
    status = _capi.ERR_NO_DEVICE


_BY_STATUS = {c.status: c for c in (InvalidDataException, InvalidOperationException, NotSupportedException,
                                    ArgumentException, DeviceError, NoDeviceError)}
_BY_STATUS[_capi.ERR_OOM] = DeviceError


def raise_for_status(status, message):
    if status == _capi.OK:
        return
    if isinstance(message, bytes):
        message = message.decode("utf-8", "replace")
    raise _BY_STATUS.get(status, JpegError)(message or _capi.lib.jpgpu_status_string(status).decode())
