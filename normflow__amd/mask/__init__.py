from .mask import Mask, EvenOddMask, AlongAxesEvenOddMask, DummyMask
