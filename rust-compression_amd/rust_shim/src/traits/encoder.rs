//! `Encoder`, `EncodeExt`, `EncodeIterator` (the reference's `src/traits/encoder.rs:12-93`): a pull
//! based byte pipeline.  `iter.encode(&mut encoder, action)` yields `Result<Out, Error>` items; the
//! encoder is borrowed for the adaptor's lifetime and can be driven again afterwards (that is how
//! `Action::Run` / `Flush` sequences are written).
use crate::action::Action;
use crate::error::CompressionError;

pub trait Encoder
where
    CompressionError: From<Self::Error>,
{
    type Error;
    type In;
    type Out;
    /// One poll: pulls input from `iter` as needed; `None` when nothing more comes out for this
    /// `action` with `iter` exhausted.
    fn next<I: Iterator<Item = Self::In>>(
        &mut self,
        iter: &mut I,
        action: Action,
    ) -> Option<Result<Self::Out, Self::Error>>;
}

pub trait EncodeExt<I>
where
    I: Iterator,
{
    fn encode<E: Encoder<In = I::Item>>(self, encoder: &mut E, action: Action) -> EncodeIterator<'_, I, E>
    where
        CompressionError: From<E::Error>;
}

impl<I> EncodeExt<I::IntoIter> for I
where
    I: IntoIterator,
{
    fn encode<E: Encoder<In = I::Item>>(self, encoder: &mut E, action: Action) -> EncodeIterator<'_, I::IntoIter, E>
    where
        CompressionError: From<E::Error>,
    {
        EncodeIterator { encoder, action, inner: self.into_iter() }
    }
}

#[derive(Debug)]
pub struct EncodeIterator<'a, I, E>
where
    I: Iterator<Item = E::In>,
    E: Encoder,
    CompressionError: From<E::Error>,
{
    encoder: &'a mut E,
    action: Action,
    inner: I,
}

impl<I, E> Iterator for EncodeIterator<'_, I, E>
where
    I: Iterator<Item = E::In>,
    E: Encoder,
    CompressionError: From<E::Error>,
{
    type Item = Result<E::Out, E::Error>;

    fn next(&mut self) -> Option<Self::Item> {
        self.encoder.next(&mut self.inner, self.action)
    }
}
