//! `Action` (the reference's `src/action.rs:8-13`): what an `Encoder::next` call does when its input
//! iterator is exhausted.

#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash)]
pub enum Action {
    /// keep everything pending; more input will follow
    Run,
    /// write out what is pending (the codecs' Flush semantics differ: see each encoder)
    Flush,
    /// end of the stream
    Finish,
}

impl Action {
    /// the C ABI's BZ_ACTION_* value
    pub(crate) fn code(self) -> i32 {
        match self {
            Action::Run => 0,
            Action::Flush => 1,
            Action::Finish => 2,
        }
    }
}
