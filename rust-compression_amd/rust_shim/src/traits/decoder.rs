//! `Decoder`, `DecodeExt`, `DecodeIterator` (the reference's `src/traits/decoder.rs:15-99`).  The
//! reference's `DecodeIterator` carries a third type parameter for the way it holds the input
//! (`BorrowMut<I>`); code that only calls `.decode(&mut d)` does not see the difference.
use crate::error::CompressionError;

pub trait Decoder {
    type Error;
    type Input;
    type Output;
    fn next<I: Iterator<Item = Self::Input>>(&mut self, iter: &mut I) -> Option<Result<Self::Output, Self::Error>>;
}

pub trait DecodeExt<I>
where
    I: Iterator,
{
    fn decode<D: Decoder<Input = I::Item>>(self, decoder: &mut D) -> DecodeIterator<'_, I, D>
    where
        CompressionError: From<D::Error>;
}

impl<I> DecodeExt<I::IntoIter> for I
where
    I: IntoIterator,
{
    fn decode<D: Decoder<Input = I::Item>>(self, decoder: &mut D) -> DecodeIterator<'_, I::IntoIter, D>
    where
        CompressionError: From<D::Error>,
    {
        DecodeIterator { decoder, inner: self.into_iter() }
    }
}

pub struct DecodeIterator<'a, I, D>
where
    I: Iterator<Item = D::Input>,
    D: Decoder,
    CompressionError: From<D::Error>,
{
    decoder: &'a mut D,
    inner: I,
}

impl<I, D> Iterator for DecodeIterator<'_, I, D>
where
    I: Iterator<Item = D::Input>,
    D: Decoder,
    CompressionError: From<D::Error>,
{
    type Item = Result<D::Output, D::Error>;

    fn next(&mut self) -> Option<Self::Item> {
        self.decoder.next(&mut self.inner)
    }
}
